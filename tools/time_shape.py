"""Wall time of forward-backward + gradient at an ad-hoc shape (GPU box):
   python tools/time_shape.py L D W T U [precision=1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import Case
L, D, W, T, U = (int(x) for x in sys.argv[1:6])
prec = int(sys.argv[6]) if len(sys.argv) > 6 else 1
c = Case(L=L, D=D, in_w=W, Ts=[T] * U, seed=1, lam_scale=0.02, precision=prec, scratch_bytes=64 << 30)
eng = c.engine(); b = c.batch(eng)
eng.fb_batch(b)
eng.enable_timing(True)
t0 = time.perf_counter()
for _ in range(3):
    eng.zero_grad(); eng.fb_batch(b)
dt = (time.perf_counter() - t0) / 3
print("L=%d D=%d W=%d T=%d U=%d prec=%d fused=%s: %.2f ms, %.0f utt/s" % (L, D, W, T, U, prec, eng.batch_is_fused(b), 1e3 * dt, U / dt),
      {k: round(v[0], 2) for k, v in eng.last_timing().items() if v[0] > 0})

"""Times one forward-backward + gradient pass at a named BASELINE shape (for rocprofv3 runs):
   python tools/prof_shape.py cfg3x64|cfg5x32 [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from cases import Case
SHAPES = {"cfg3x64": dict(L=48, D=10, in_w=144, Ts=[304] * 64, trans_ctx=6, seed=4, lam_scale=0.01),
          "cfg3x256": dict(L=48, D=10, in_w=144, Ts=[304] * 256, trans_ctx=6, seed=4, lam_scale=0.01),
          "cfg5x32": dict(L=200, D=40, in_w=123, Ts=[2000] * 32, seed=6, lam_scale=0.01),
          "cfg5x128": dict(L=200, D=40, in_w=123, Ts=[2000] * 128, seed=6, lam_scale=0.01)}
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
c = Case(precision=int(os.environ.get("PREC", "3")), scratch_bytes=160 << 30, **SHAPES[name])
eng = c.engine(); b = c.batch(eng)
eng.fb_batch(b)
eng.enable_timing(True)
for _ in range(reps):
    eng.zero_grad(); eng.fb_batch(b)
print(name, {k: round(v[0], 2) for k, v in eng.last_timing().items()})
for nm, ms, nl in sorted(eng.kernel_timing(), key=lambda x: -x[1])[:12]:
    print("   %-44s %9.3f ms  (%d launches)" % (nm, ms, nl))

"""Random shapes through the fused kernels (GPU box): FAST gradient with fused window synthesis against
the materialised-window kernels (SCRF_FUSE=0) and, for small cases, against the oracle; fast decode
against the EXACT decode.  usage: python tools/fused_shape_sweep.py [n_shapes] [seed]
SWEEP_PREC=3 runs the fused side under FASTLIN (bounds 1e-6 / 1e-8: its window average is the exact mean);
SWEEP_LONG=1 appends a 260-frame utterance to every case (launches of few utterances then walk the posterior pass in
segments: k_post_z's split form)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from cases import Case

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
PREC = int(os.environ.get("SWEEP_PREC", "1"))
LONG = os.environ.get("SWEEP_LONG", "0") == "1"
TOL_G, TOL_Z, TOL_O = (1e-9, 1e-11, 1e-8) if PREC == 1 else (1e-6, 1e-8, 1e-6)
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
def one(kw, D, W, L, Ts):
    res = {}
    fused = None
    for fuse in ("1", "0"):
        os.environ["SCRF_FUSE"] = fuse
        c = Case(precision=PREC if fuse == "1" else 1, **kw)
        eng = c.engine(); b = c.batch(eng)
        if fuse == "1":
            fused = eng.batch_is_fused(b)
        numer, zx = eng.fb_batch(b)
        g = eng.get_grad()
        labs, cost = eng.viterbi_batch(b)
        res[fuse] = (numer, zx, g, labs, cost, eng.decode_stats())
        b.close(); eng.close()
    os.environ["SCRF_FAST_DECODE"] = "0"
    c = Case(precision=1, **kw); eng = c.engine(); b = c.batch(eng)
    elabs, ecost = eng.viterbi_batch(b)
    b.close(); eng.close()
    del os.environ["SCRF_FAST_DECODE"]
    (n1, z1, g1, l1, c1, st), (n0, z0, g0, l0, c0, _) = res["1"], res["0"]
    scale = max(np.abs(g0).max(), 1e-300)
    e_g = np.abs(g1 - g0).max() / scale
    e_z = np.abs(z1 - z0).max() / max(1.0, np.abs(z0).max())
    ok_dec = all(list(a) == list(b_) for a, b_ in zip(l1, elabs)) and c1.tobytes() == ecost.tobytes()
    ok = e_g < TOL_G and e_z < TOL_Z and ok_dec
    og_err = -1.0
    if sum(Ts) * D * L * (8 * W + D) < 4e8:
        og, on, oz = c.oracle_gradient()
        og_err = np.abs(g1 - og).max() / max(np.abs(og).max(), 1e-300)
        ok = ok and og_err < TOL_O
    print("%s D=%d W=%d L=%d Ts=%s fused=%s grad_vs_mat=%.1e zx=%.1e oracle=%.1e decode_equal=%s fix=%s" %
          ("ok  " if ok else "FAIL", D, W, L, Ts, fused, e_g, e_z, og_err, ok_dec, st), flush=True)
    return ok


for i in range(n):
    D = int(rng.choice([2, 3, 4, 5, 7, 10, 12, 13, 16, 25, 26, 32, 40]))
    W = int(rng.choice([1, 2, 3, 5, 8, 13, 16, 17, 39, 40, 41, 64, 69]))
    L = int(rng.choice([2, 3, 7, 16, 47, 48, 49, 64, 65, 96, 130]))
    Ts = [int(x) for x in rng.choice([1, 2, D - 1, D, D + 1, 2 * D + 3, 57, 130], size=int(rng.randint(1, 6)))]
    Ts = [max(1, t) for t in Ts] + ([260] if LONG else [])
    kw = dict(L=L, D=D, in_w=W, Ts=Ts, seed=1000 + i, lam_scale=0.2)
    try:
        ok = one(kw, D, W, L, Ts)
    except Exception as e:
        print("FAIL D=%d W=%d L=%d Ts=%s exception %s" % (D, W, L, Ts, str(e)[:200]), flush=True)
        for k in ("SCRF_FAST_DECODE",):
            os.environ.pop(k, None)
        ok = False
    bad += 0 if ok else 1
print("failures:", bad)
sys.exit(1 if bad else 0)

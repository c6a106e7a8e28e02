"""One-off stress (GPU box): many short utterances over many chunks, and very long utterances,
FAST vs EXACT gradients and fast vs exact decode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from cases import Case
rng = np.random.RandomState(0)
bad = 0
for name, kw, scratch in [
        ("many-short", dict(L=48, D=10, in_w=13, Ts=[int(t) for t in rng.randint(1, 60, 3000)], seed=1, lam_scale=0.05), 8 << 20),
        ("long", dict(L=48, D=25, in_w=39, Ts=[5000, 1, 7000, 300], seed=2, lam_scale=0.02), 0),
        ("long-L100", dict(L=100, D=10, in_w=8, Ts=[4000, 33], seed=3, lam_scale=0.02), 0),
        ("long-hybrid", dict(L=200, D=40, in_w=60, Ts=[2500, 50, 700], seed=4, lam_scale=0.01), 0),       # round 4: hybrid path, k_lin_z5 segments
        ("few-long-48", dict(L=48, D=25, in_w=39, Ts=[3000, 2999], seed=5, lam_scale=0.02), 0)]:          # round 4: k_post_z in segments
    res = {}
    for prec in (0, 1):
        c = Case(precision=prec, scratch_bytes=scratch, **kw)
        eng = c.engine(); b = c.batch(eng)
        numer, zx = eng.fb_batch(b); g = eng.get_grad()
        res[prec] = (numer, zx, g)
        if prec == 1:
            labs, cost = eng.viterbi_batch(b); st = eng.decode_stats()
        b.close(); eng.close()
    os.environ["SCRF_FAST_DECODE"] = "0"
    c = Case(precision=0, scratch_bytes=scratch, **kw); eng = c.engine(); b = c.batch(eng)
    elabs, ecost = eng.viterbi_batch(b); b.close(); eng.close()
    del os.environ["SCRF_FAST_DECODE"]
    (n0, z0, g0), (n1, z1, g1) = res[0], res[1]
    e_g = np.abs(g1 - g0).max() / np.abs(g0).max()
    e_z = np.abs(z1 - z0).max() / np.abs(z0).max()
    dec = all(list(a) == list(b_) for a, b_ in zip(labs, elabs)) and cost.tobytes() == ecost.tobytes()
    ok = e_g < 1e-9 and e_z < 1e-11 and dec
    bad += 0 if ok else 1
    print("%s %s: grad FAST vs EXACT %.1e, zx %.1e, decode equal %s, fix-ups %s" % ("ok  " if ok else "FAIL", name, e_g, e_z, dec, st), flush=True)
sys.exit(1 if bad else 0)

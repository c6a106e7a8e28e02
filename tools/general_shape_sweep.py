"""Random shapes through every training / decode path (GPU box): stdstate and stdtrans maps (context
stream), EXACT / FAST / FAST32, tiny scratch budgets (many chunks), L up to 300 (multi-wavefront and
generic recursions), D up to 45, wide streams.  Each case against the oracle (gradient, Zx, numerator;
Viterbi labels and cost on one utterance).  usage: python tools/general_shape_sweep.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc
from cases import Case

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
TOL = {0: (1e-10, 1e-12), 1: (1e-9, 1e-11), 2: (5e-5, 1e-5)}   # gradient (relative to its max), Zx and numerator; contract 1e-4


def one(kw, prec, scratch):
    c = Case(precision=prec, scratch_bytes=scratch, **kw)
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    labs, cost = eng.viterbi_batch(b)
    og, on, oz = c.oracle_gradient()
    e_g = np.abs(g - og).max() / max(np.abs(og).max(), 1e-300)
    e_z = np.abs(zx - oz).max() / max(1.0, np.abs(oz).max())
    e_n = np.abs(numer - on).max() / max(1.0, np.abs(on).max())
    u = len(c.Ts) - 1; T = c.Ts[u]
    So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
    if c.ocfg.model_type == orc.STDFRAME:
        oa, ons, ofin = orc.frame_lattice_arcs(c.ocfg, So, Mo, T)
    else:
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
    ol, oc = orc.best_path(oa, ons, ofin)
    dec = list(labs[u]) == list(ol) and np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()
    if len(oa) < 3000000:   # lattice arcs in AddArc order, bit for bit
        ga, gns, gfin = eng.lattice_arcs(b, u)
        dec = dec and (gns, gfin) == (ons, ofin) and ga.tobytes() == oa.tobytes()
    b.close(); eng.close()
    ok = e_g <= TOL[prec][0] and e_z <= TOL[prec][1] and e_n <= TOL[prec][1] and dec
    return ok, "grad=%.1e zx=%.1e numer=%.1e decode=%s" % (e_g, e_z, e_n, dec)


bad = 0
for i in range(n):
    D = int(rng.choice([2, 3, 5, 8, 10, 17, 25, 33, 40, 45]))
    W = int(rng.choice([1, 2, 4, 9, 16, 39, 70, 90]))
    L = int(rng.choice([2, 5, 16, 48, 50, 64, 65, 100, 200, 256, 257, 300]))
    ctx = rng.choice([-1, -1, 1, 3])
    prec = int(rng.choice([0, 1, 1, 2]))
    nu = int(rng.randint(1, 5))
    Ts = [max(1, int(x)) for x in rng.choice([1, 2, D - 1, D, D + 2, 3 * D, 40], size=nu)]
    # keep the oracle affordable
    cost = sum(Ts) * D * L * (8 * W + D) + (sum(Ts) * L * L * (2 * ctx + 1) * W if ctx > 0 else 0)
    if cost > 3e8:
        L = min(L, 50); W = min(W, 16)
    scratch = int(rng.choice([0, 0, 1 << 16, 1 << 20]))
    frame = rng.rand() < 0.2
    if frame:   # frame-level CRF (BASELINE config 1): D = 1, transition features optional
        D = 1
        Ts = [max(1, int(x)) for x in rng.choice([1, 2, 3, 9, 40], size=nu)]
    kw = dict(L=L, D=D, in_w=W, Ts=Ts, seed=2000 + i, lam_scale=0.05, trans_ctx=(int(ctx) if ctx > 0 else (0 if frame else None)),
              frame_model=bool(frame))
    tag = ("frame " if frame else "") + "D=%d W=%d L=%d ctx=%s prec=%d scratch=%d Ts=%s" % (D, W, L, kw["trans_ctx"], prec, scratch, Ts)
    try:
        ok, msg = one(kw, prec, scratch)
    except Exception as e:
        # L > 256 or D > 40 with 2 * D * L doubles beyond the LDS is refused with a message, not a launch failure
        refused = "too large for the workgroup-per-utterance recursion" in str(e) and (L > 256 or D > 40) and 16 * D * L > 150 * 1024
        ok, msg = refused, ("refused (shape beyond the generic kernel)" if refused else "exception " + str(e)[:300])
    bad += 0 if ok else 1
    print(("ok   " if ok else "FAIL ") + tag + " " + msg, flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)

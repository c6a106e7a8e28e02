"""Runs the engine on the other BASELINE shapes (config 3: TIMIT demo with transition features,
config 5: stress L=200 D=40) at reduced batch, checks one utterance against the oracle and prints
phase timings.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, scrf_amd
from cases import Case

def run(name, prec, **kw):
    t0 = time.time()
    c = Case(precision=prec, **kw)
    eng = c.engine(); b = c.batch(eng)
    eng.enable_timing(True)
    numer, zx = eng.fb_batch(b)
    tm = eng.last_timing()
    g = eng.get_grad()
    # oracle on utterance 0 only
    cfg, lay = c.ocfg, c.olay
    t1 = time.time()
    rc, og, on, oz = orc.seg_build_gradient(cfg, lay, c.lam, c.windows(0), c.labels[0], c.Ts[0])
    t2 = time.time()
    eng.zero_grad()
    b1 = eng.batch_from_frames(c.frames[:1], c.labels[:1], c.recipes, [c.frames2[:1]] if c.frames2 is not None else None)
    n1, z1 = eng.fb_batch(b1); g1 = eng.get_grad()
    err = np.abs(g1 - og).max() / np.abs(og).max()
    labs, cost = eng.viterbi_batch(b1)
    S, M = orc.seg_scores(cfg, lay, c.lam, c.windows(0), c.Ts[0])
    arcs, ns, fin = orc.seg_lattice_arcs(cfg, S, M, c.Ts[0])
    ol, oc = orc.best_path(arcs, ns, fin)
    print("%s prec=%d lambda_len=%d rc=%d zx_rel=%.2e numer_rel=%.2e grad_rel=%.2e viterbi_equal=%s cost_equal=%s oracle_s=%.1f"
          % (name, prec, lay.lambda_len, rc, abs(z1[0] - oz) / abs(oz), abs(n1[0] - on) / max(1, abs(on)), err,
             list(labs[0]) == list(ol), np.float32(cost[0]).tobytes() == np.float32(oc).tobytes(), t2 - t1), flush=True)
    print("   phases(ms) for %d utts:" % len(c.Ts), {k: round(v[0], 2) for k, v in tm.items()}, "setup %.1fs" % (t1 - t0), flush=True)
    b.close(); b1.close(); eng.close()

if __name__ == "__main__":
    # config 3 shape: L=48 D=10, 144-dim frames, +-6 context transition features (lambda_len 4,371,216)
    for prec in (0, 1):
        run("cfg3", prec, L=48, D=10, in_w=144, Ts=[120, 304, 200, 260], trans_ctx=6, seed=3, lam_scale=0.01)
    # config 3 shape at a minibatch that fills the chip: 64 utterances of ~300 frames
    run("cfg3x64", 1, L=48, D=10, in_w=144, Ts=[304] * 64, trans_ctx=6, seed=4, lam_scale=0.01)
    # config 5 shape: L=200 D=40, 123-dim frames (generic workgroup-per-utterance DP kernel)
    run("cfg5x32", 1, L=200, D=40, in_w=123, Ts=[2000] * 32, seed=6, lam_scale=0.01)
    for prec in (0, 1):
        run("cfg5", prec, L=200, D=40, in_w=123, Ts=[300, 500], seed=5, lam_scale=0.01)

"""Random shapes through the secondary model types' kernels (STDSEG_NO_DUR, STDSEG, the n-state frame model) against the
oracle: gradient, numerator, Zx, lattice arcs, best path.  python tools/secondary_shape_sweep.py [n] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, scrf_amd
from cases import Case

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
fails = 0
for it in range(n):
    kind = ["segtrans", "stdseg", "nstate"][it % 3]
    try:
        if kind == "nstate":
            K = int(rng.randint(2, 5)); P = int(rng.choice([1, 2, 3, 15, 16, 17, 33, 63, 64, 65])); F = int(rng.randint(1, 5))
            Ts = [int(rng.randint(1, 12)) for _ in range(int(rng.randint(1, 4)))]
            L = P * K
            cfgkw = dict(model_type=orc.STDFRAME, L=L, D=1, F=F, use_trans_ftrs=bool(rng.randint(0, 2)), tfs=0, tfe=F - 1, num_states=K)
            ocfg = orc.config(**cfgkw); olay = orc.Layout(ocfg)
            lam = rng.normal(0, 0.2, olay.lambda_len)
            frames = [rng.random_sample((T, F)).astype(np.float32) for T in Ts]
            labels = []
            for T in Ts:
                lab = np.zeros(T, dtype=np.uint32); c = int(rng.randint(0, L))
                for t in range(T):
                    lab[t] = c
                    if rng.rand() >= 0.4:
                        c = int(rng.randint(0, P)) * K if (c + 1) % K == 0 else c + 1
                labels.append(lab)
            eng = scrf_amd.Engine(scrf_amd.make_config(precision=int(rng.randint(0, 2)), **cfgkw)); eng.set_lambda(lam)
            b = eng.batch_from_frames(frames, labels, [scrf_amd.StreamRecipe(F, 0, 0, 0)], None)
            numer, zx = eng.fb_batch(b); g = eng.get_grad()
            og = np.zeros(olay.lambda_len); on = []; oz = []
            for u, T in enumerate(Ts):
                rc, og, nn, zz = orc.nstate_build_gradient(ocfg, olay, lam, frames[u], labels[u], T, grad=og)
                assert rc == 0
                on.append(nn); oz.append(zz)
            desc = "P=%d K=%d F=%d Ts=%s" % (P, K, F, Ts)
            labs, cost = eng.viterbi_batch(b)
            for u, T in enumerate(Ts):
                S, TD, TO, TE = orc.nstate_scores(ocfg, olay, lam, frames[u], T)
                oa, ons, ofin = orc.nstate_lattice_arcs(ocfg, S, TD, TO, TE, T)
                ga, gns, gfin = eng.lattice_arcs(b, u)
                assert ga.tobytes() == oa.tobytes(), "arcs"
                ol, oc = orc.best_path(oa, ons, ofin)
                assert list(labs[u]) == list(ol), "best path"
        else:
            mt = orc.STDSEG_NO_DUR if kind == "segtrans" else orc.STDSEG
            L = int(rng.choice([1, 2, 3, 5, 15, 16, 17, 31, 48, 63, 64, 65, 70])) if kind == "segtrans" else int(rng.choice([1, 2, 3, 7, 16, 33, 64, 65]))
            D = int(rng.randint(1, 21)) if kind == "segtrans" else int(rng.randint(1, 6))
            W = int(rng.randint(1, 4))
            Ts = [int(rng.randint(1, 2 * D + 6)) for _ in range(int(rng.randint(1, 4)))]
            kw = dict(L=L, D=D, in_w=W, Ts=Ts, seed=int(rng.randint(1 << 30)), model_type=mt, precision=int(rng.randint(0, 2)), lam_scale=0.2)
            Fs = 8 * W + D
            if rng.randint(0, 2) and D > 1:
                lo = int(rng.randint(0, Fs - 1)); kw["trans_share"] = (lo, int(rng.randint(lo, min(Fs - 1, lo + 6) + 1)))
            if kind == "stdseg" and L * D > 260:
                D = max(1, 260 // L); kw["D"] = D; kw.pop("trans_share", None); kw["Ts"] = Ts = [min(T, 3 * D + 2) for T in Ts]
            c = Case(**kw)
            eng = c.engine(); b = c.batch(eng)
            numer, zx = eng.fb_batch(b); g = eng.get_grad()
            og, on, oz = c.oracle_gradient()
            desc = "L=%d D=%d W=%d Ts=%s share=%s" % (L, D, W, Ts, kw.get("trans_share"))
            labs, cost = eng.viterbi_batch(b)
            for u, T in enumerate(c.Ts):
                if kind == "segtrans":
                    S, M = orc.segtrans_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
                    oa, ons, ofin = orc.segtrans_lattice_arcs(c.ocfg, S, M, T)
                else:
                    S, M = orc.stdseg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
                    oa, ons, ofin = orc.stdseg_lattice_arcs(c.ocfg, S, M, T)
                ga, gns, gfin = eng.lattice_arcs(b, u)
                assert ga.tobytes() == oa.tobytes(), "arcs"
                ol, oc = orc.best_path(oa, ons, ofin)
                assert list(labs[u]) == list(ol), "best path"
        on = np.array(on); oz = np.array(oz)
        e_n = np.abs(numer - on).max() / max(1, np.abs(on).max()); e_z = np.abs(zx - oz).max() / max(1, np.abs(oz).max())
        e_g = np.abs(g - og).max() / max(1e-6, np.abs(og).max())   # L = 1: the gradient is 0 up to rounding
        assert e_n <= 1e-9 and e_z <= 1e-9 and e_g <= 1e-8, (e_n, e_z, e_g)
        print("ok   %-8s %s  grad %.1e zx %.1e" % (kind, desc, e_g, e_z), flush=True)
        b.close(); eng.close()
    except Exception as ex:  # noqa: BLE001
        fails += 1
        print("FAIL %-8s %s: %r" % (kind, locals().get("desc", "?"), ex), flush=True)
print("failures:", fails)
sys.exit(1 if fails else 0)

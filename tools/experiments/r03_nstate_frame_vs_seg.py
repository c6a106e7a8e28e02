"""experiment: is the n-state FRAME model (STDFRAME, K states per phone) the same function as the n-state segmental
model with D = 1 (the shadow layout over the dense kernels)?  gradient, numerator, Zx of both engines on the same data"""
import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import orc, scrf_amd
rng = np.random.RandomState(5)
for (P, K, F, U, T, tf, prec) in [(3, 3, 4, 5, 9, False, 0), (16, 3, 6, 8, 20, False, 0), (16, 3, 6, 8, 20, True, 0), (48, 3, 39, 256, 300, False, 1)]:
    L = P * K
    frames = [rng.random_sample((T, F)).astype(np.float32) for _ in range(U)]
    labels = []
    for _ in range(U):
        labs = np.zeros(T, dtype=np.uint32); c = int(rng.randint(0, P)) * K
        for t in range(T):
            labs[t] = c
            if rng.rand() >= 0.4:
                c = int(rng.randint(0, P)) * K if (c + 1) % K == 0 else c + 1
        # end in an end state (the segmental model demands it)
        labels.append(labs)
    kw = dict(L=L, D=1, F=F, use_trans_ftrs=tf, tfs=0, tfe=F - 1, num_states=K)
    res = []
    for mt in (orc.STDFRAME, orc.STDSEG_NO_DUR_NO_SEGTRANSFTR):
        try:
            eng = scrf_amd.Engine(scrf_amd.make_config(model_type=mt, precision=prec, scratch_bytes=32 << 30, **kw))
        except Exception as e:
            print("  ", mt, "refused:", e); res.append(None); continue
        if not res or res[0] is None:
            lam = rng.normal(0, 0.1, eng.lambda_len)
        eng.set_lambda(lam)
        b = eng.batch_from_frames(frames, labels, [scrf_amd.StreamRecipe(F, 0, 0, 0)], None)
        eng.zero_grad()
        try:
            n, z = eng.fb_batch(b)
        except Exception as e:
            print("  ", mt, "fb_batch:", e); res.append(None); continue
        g = eng.get_grad().copy()
        eng.synchronize(); t0 = time.perf_counter()
        for _ in range(3):
            eng.zero_grad(); eng.fb_batch(b, want_scalars=False)
        eng.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 3
        res.append((eng.lambda_len, n, z, g, ms))
        b.close(); eng.close()
    if res[0] and res[1]:
        a, c = res
        print("P=%d K=%d F=%d U=%d T=%d trans_ftrs=%s prec=%d: lambda_len %d / %d, ms %.2f / %.2f" % (P, K, F, U, T, tf, prec, a[0], c[0], a[4], c[4]))
        if a[0] == c[0]:
            print("   zx rel %.3g  numer rel %.3g  grad rel %.3g" % (np.abs(a[2] - c[2]).max() / np.abs(a[2]).max(), np.abs(a[1] - c[1]).max() / max(1, np.abs(a[1]).max()), np.abs(a[3] - c[3]).max() / max(1e-300, np.abs(a[3]).max())))

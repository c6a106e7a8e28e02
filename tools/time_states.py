"""Step time of the config-2 shape with K states per phone (crf_states = K on stdseg_no_dur_no_segtransftr) next to the
one-state model of the same label count: the masked model runs the same kernels.  python tools/time_states.py [utts]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "asr-craft_amd", "python"))
import numpy as np
import scrf_amd

U = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L, D, W, T = 48, 25, 39, 300
rng = np.random.RandomState(0)
frames = [rng.random_sample((T, W)).astype(np.float32) for _ in range(U)]
for K in (1, 3):
    labels = []
    for _ in range(U):   # segments of 5 frames, labels walking the topology
        lab = np.full(T, 0xffffffff, dtype=np.uint32)
        c = int(rng.randint(0, L))
        for t in range(4, T, 5):
            lab[t] = c + L * 4
            c = (int(rng.randint(0, L // K)) * K) if (c + 1) % K == 0 else c + 1
        labels.append(lab)
    cfg = scrf_amd.make_config(model_type=scrf_amd.STDSEG_NO_DUR_NO_SEGTRANSFTR, L=L, D=D, F=8 * W + D, precision=1, num_states=K, scratch_bytes=64 << 30)
    eng = scrf_amd.Engine(cfg)
    eng.set_lambda(rng.normal(0, 0.05, eng.lambda_len))
    b = eng.batch_from_frames(frames, labels, [scrf_amd.StreamRecipe(W, 0, 0, 1)], None)
    for _ in range(2):
        eng.fb_batch(b, want_scalars=False); eng.sgd_step(1e-4, False)
    eng.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        eng.fb_batch(b, want_scalars=False); eng.sgd_step(1e-4, False)
    eng.synchronize()
    dt = (time.time() - t0) / n
    print("K=%d lambda_len=%d  %.2f ms/step  %.1f k utterances/s  log-domain redos %d" % (K, eng.lambda_len, dt * 1e3, U / dt / 1e3, eng.train_stats()), flush=True)
    b.close(); eng.close()

"""Forward-backward + gradient step time of every model type at a TIMIT-like shape (48 phones, durations <= 10,
39-dim x 300-frame synthetic utterances).  python tools/time_models.py [utts]"""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "asr-craft_amd", "python"))
import numpy as np
import scrf_amd
from scrf_amd import synth

U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
only = sys.argv[2] if len(sys.argv) > 2 else None
P, D, W, T = 48, 10, 39, 300
rng = np.random.RandomState(0)
frames = [rng.random_sample((T, W)).astype(np.float32) for _ in range(U)]
Fs = 8 * W + D
MODELS = [
    ("stdseg_no_dur_no_segtransftr", dict(model_type=scrf_amd.STDSEG_NO_DUR_NO_SEGTRANSFTR, L=P, D=D, F=Fs), P, D),
    ("stdseg_no_dur (transition features = mean block of the window)", dict(model_type=scrf_amd.STDSEG_NO_DUR, L=P, D=D, F=Fs, use_trans_ftrs=True, tfs=0, tfe=W - 1), P, D),
    ("stdseg (labels carry the duration)", dict(model_type=scrf_amd.STDSEG, L=P * D, D=D, F=Fs), P, D),
    ("stdframe, 3 states per phone", dict(model_type=scrf_amd.STDFRAME, L=P * 3, D=1, F=W, num_states=3), P * 3, 1),
    ("stdframe, 3 states per phone, as CRFTrain runs it (masked dense layout, maximum duration 1)", dict(model_type=scrf_amd.STDSEG_NO_DUR_NO_SEGTRANSFTR, L=P * 3, D=1, F=W, num_states=3), P * 3, 1),
    ("stdframe", dict(model_type=scrf_amd.STDFRAME, L=P, D=1, F=W), P, 1),
]
for name, kw, L, Dm in MODELS:
    if only and only not in name:
        continue
    labels = [synth.group_labels(synth.frame_labels(rng, T, L, Dm), Dm, L) if Dm > 1 else rng.randint(0, L, T).astype(np.uint32) for _ in range(U)]
    if kw.get("num_states", 1) > 1:
        labels = [np.minimum(np.arange(T) // 4 % 3 + 3 * (np.arange(T) // 12 % P), L - 1).astype(np.uint32) for _ in range(U)]
    cfg = scrf_amd.make_config(precision=1, scratch_bytes=32 << 30, **kw)
    eng = scrf_amd.Engine(cfg)
    eng.set_lambda(rng.normal(0, 0.02, eng.lambda_len))
    b = eng.batch_from_frames(frames, labels, [scrf_amd.StreamRecipe(W, 0, 0, 1 if Dm > 1 else 0)], None)
    eng.fb_batch(b, want_scalars=False); eng.sgd_step(1e-5, False); eng.synchronize()
    t0 = time.time()
    n = 3
    for _ in range(n):
        eng.fb_batch(b, want_scalars=False); eng.sgd_step(1e-5, False)
    eng.synchronize()
    dt = (time.time() - t0) / n
    eng.enable_timing(True)
    eng.fb_batch(b, want_scalars=False); eng.synchronize()
    kt = eng.kernel_timing()
    eng.enable_timing(False)
    print("%-70s lambda_len=%8d  %9.2f ms / %d utterances  %8.2f k utterances/s" % (name, eng.lambda_len, dt * 1e3, U, U / dt / 1e3), flush=True)
    for nm, ms, nl in sorted(kt, key=lambda x: -x[1])[:8]:
        print("      %-28s %10.3f ms  (%d launches)" % (nm, ms, nl), flush=True)
    b.close(); eng.close()

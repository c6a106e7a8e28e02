"""Synthetic utterances of the BASELINE.json shapes.

RNG = MT19937 (numpy RandomState) seeded per utterance from a base seed (1234), frame
features float32 ~ U[0,1), labels from a random segmentation whose segment lengths are
1 + rng % (2D) (so some exceed D and exercise the reference's over-long-segment
splitting, io/CRF_InLabStream_SeqMultiWindow.cpp:51-110), phone = rng % L, weights
lambda ~ N(0, 0.01^2) fp64.  There is no network for TIMIT, so this is what bench.py
and the tests feed the engine.
"""
import numpy as np

LAB_BAD = 0xFFFFFFFF


def frame_labels(rng, T, L, D):
    """Random frame-level phone labels: runs of length 1 + rng % (2D)."""
    labs = np.empty(T, dtype=np.uint32)
    t = 0
    prev = -1
    while t < T:
        n = 1 + int(rng.randint(0, 2 * D))
        ph = int(rng.randint(0, L))
        if ph == prev:  # identical neighbours would merge into one run
            ph = (ph + 1) % L if L > 1 else ph
        labs[t:t + n] = ph
        prev = ph
        t += n
    return labs


def group_labels(frame_labs, D, L):
    """Frame labels -> per-end-frame segment label L*(dur-1)+phone or LAB_BAD, splitting runs
    longer than D evenly (io/CRF_InLabStream_SeqMultiWindow.cpp:51-110; label id as in
    trainers/gradbuilders/CRF_NewGradBuilder_StdSeg_NoDur_NoTrans.cpp:216-231)."""
    T = len(frame_labs)
    out = np.full(T, LAB_BAD, dtype=np.uint32)
    start = 0
    while start < T:
        lab = int(frame_labs[start])
        nxt = start + 1
        while nxt < T and int(frame_labs[nxt]) == lab:
            nxt += 1
        dur = nxt - start
        if dur <= D:
            out[nxt - 1] = L * (dur - 1) + lab
        else:
            pieces = dur // D if dur % D == 0 else dur // D + 1
            pd, rem = divmod(dur, pieces)
            ps = start
            for r in range(pieces):
                d = pd + 1 if r < rem else pd
                out[ps + d - 1] = L * (d - 1) + lab
                ps += d
        start = nxt
    return out


def make_utt(seed, T, in_width, L, D):
    rng = np.random.RandomState(seed)
    frames = rng.random_sample((T, in_width)).astype(np.float32)
    fl = frame_labels(rng, T, L, D)
    return frames, group_labels(fl, D, L)


def make_batch(U, T, in_width, L, D, seed=1234, t_jitter=0):
    """Packed batch: frames [sum T, in_width] f32, labels [sum T] u32, frame_off [U+1] u64."""
    rng = np.random.RandomState(seed)
    Ts = [T if t_jitter == 0 else int(np.clip(rng.normal(T, t_jitter), max(2, T // 5), T * 5 // 2))
          for _ in range(U)]
    off = np.zeros(U + 1, dtype=np.uint64)
    off[1:] = np.cumsum(Ts)
    frames = np.empty((int(off[-1]), in_width), dtype=np.float32)
    labels = np.empty(int(off[-1]), dtype=np.uint32)
    for u in range(U):
        f, l = make_utt(seed + 1 + u, Ts[u], in_width, L, D)
        frames[int(off[u]):int(off[u + 1])] = f
        labels[int(off[u]):int(off[u + 1])] = l
    return frames, labels, off


def make_lambda(n, seed=1234, scale=0.01):
    rng = np.random.RandomState(seed ^ 0x5EED)
    return rng.normal(0.0, scale, size=n).astype(np.float64)

"""Shared construction of test cases: the same seeded inputs for the oracle and the engine."""
import numpy as np

import orc
import scrf_amd
from scrf_amd import synth


def nstate_segment_labels(rng, L, K, T, D):
    """A random segmentation whose labels follow the K-states-per-phone topology (stay, advance, or end state -> a
    start state), in the segment-end labelling of the label stream: label + L*(dur-1) on a segment's last frame."""
    out = np.full(T, 0xffffffff, dtype=np.uint32)
    c = int(rng.randint(0, L))
    t = -1
    while t + 1 < T:
        d = int(rng.randint(1, min(D, T - 1 - t) + 1))
        t += d
        out[t] = c + L * (d - 1)
        if rng.rand() >= 0.3:
            c = int(rng.randint(0, L // K)) * K if (c + 1) % K == 0 else c + 1
    return out


class Case:
    """L labels, max duration D, raw frame width in_w, utterance lengths Ts.
    trans_ctx=None: `stdstate` map (bias-only transitions); trans_ctx=c: `stdtrans` map whose
    transition features are a second stream of boundary context (c frames each side), like the
    TIMIT demo's ftr2 stream (demo/segmental-timit-demo.cfg.in:21-24)."""

    def __init__(self, L, D, in_w, Ts, trans_ctx=None, seed=0, lam_scale=0.3, frame_model=False,
                 scratch_bytes=0, precision=0, l1_norm=False, model_type=None, trans_share=None,
                 num_states=1, conform_labels=True, exact_avg=False):
        self.L, self.D, self.in_w, self.Ts = L, D, in_w, list(Ts)
        self.trans_ctx = trans_ctx
        rng = np.random.RandomState(seed)
        self.frames = [rng.random_sample((T, in_w)).astype(np.float32) for T in Ts]
        if exact_avg:
            # frame values m * lcm(1..D) / 2^k, m in 0..4: every running float sum over <= D frames and its quotient by
            # the length is exact in float, so the reference's float window average IS the exact mean (D <= 10)
            assert D <= 10
            lcm = int(np.lcm.reduce(np.arange(1, D + 1)))
            scale = 2.0 ** -int(np.ceil(np.log2(5 * lcm)))
            self.frames = [(rng.randint(0, 5, (T, in_w)) * (lcm * scale)).astype(np.float32) for T in Ts]
        if l1_norm:   # rows L1-normalised like softmax posteriors (SURVEY 8d, config 3/4 inputs)
            self.frames = [(f / f.sum(1, keepdims=True)).astype(np.float32) for f in self.frames]
        if frame_model:
            self.labels = [rng.randint(0, L, T).astype(np.uint32) for T in Ts]
        elif num_states > 1 and conform_labels:
            self.labels = [nstate_segment_labels(rng, L, num_states, T, D) for T in Ts]
        else:
            self.labels = [synth.group_labels(synth.frame_labels(rng, T, L, D), D, L) for T in Ts]
        Fs = orc.window_width(in_w, D, 0, 0, True)
        self.recipes = [scrf_amd.StreamRecipe(in_w, 0, 0, 1)]
        self.frames2 = None
        mt = orc.STDFRAME if frame_model else orc.STDSEG_NO_DUR_NO_SEGTRANSFTR
        if model_type is not None:
            mt = model_type
        if trans_ctx is None and trans_share is not None:
            # transition features = columns [lo, hi] of the segment's own window vector (single stream)
            self.F = Fs
            kw = dict(model_type=mt, L=L, D=D, F=Fs, use_trans_ftrs=True, tfs=trans_share[0], tfe=trans_share[1])
        elif trans_ctx is None:
            self.F = Fs
            kw = dict(model_type=mt, L=L, D=D, F=Fs)
        else:
            c = trans_ctx
            Ft = orc.window_width(in_w, D, c, c, False)
            self.F = Fs + Ft
            self.frames2 = [np.concatenate([np.repeat(f[:1], c, 0), f, np.repeat(f[-1:], c, 0)]) for f in self.frames]
            self.recipes.append(scrf_amd.StreamRecipe(in_w, c, c, 0))
            kw = dict(model_type=mt, L=L, D=D, F=self.F, sfe=Fs - 1, use_trans_ftrs=True, tfs=Fs)
        self.Fs = Fs
        if num_states > 1:
            kw["num_states"] = num_states
        if mt == orc.STDSEG:   # the feature map and the weight layout run over all labels: nLabs = nActualLabs * D
            kw["L"] = L * D
        self.ocfg = orc.config(**kw)
        self.olay = orc.Layout(self.ocfg)
        self.gcfg = scrf_amd.make_config(scratch_bytes=scratch_bytes, precision=precision, **kw)
        self.lam = rng.normal(0, lam_scale, self.olay.lambda_len)

    def windows(self, u):
        """oracle window vectors of utterance u: [N_seg, F]"""
        T = self.Ts[u]
        X = np.zeros((orc.num_segs(T, self.D), self.F), dtype=np.float32)
        orc.windows(self.frames[u], self.D, 0, 0, True, out=X, out_col=0)
        if self.trans_ctx is not None:
            c = self.trans_ctx
            orc.windows(self.frames2[u], self.D, c, c, False, out=X, out_col=self.Fs)
        return X

    def engine(self):
        e = scrf_amd.Engine(self.gcfg)
        e.set_lambda(self.lam)
        return e

    def batch(self, eng, with_labels=True):
        return eng.batch_from_frames(self.frames, self.labels if with_labels else None, self.recipes,
                                     [self.frames2] if self.frames2 is not None else None)

    def oracle_gradient(self):
        """sum over utterances of buildGradient (one stream): grad, numer[], zx[]"""
        g = np.zeros(self.olay.lambda_len)
        numer, zx = [], []
        for u, T in enumerate(self.Ts):
            if self.ocfg.model_type == orc.STDFRAME:
                rc, g, n, z = orc.frame_build_gradient(self.ocfg, self.olay, self.lam, self.windows(u), self.labels[u], T, grad=g)
            elif self.ocfg.model_type == orc.STDSEG:
                rc, g, n, z = orc.stdseg_build_gradient(self.ocfg, self.olay, self.lam, self.windows(u), self.labels[u], T, grad=g)
            elif self.ocfg.model_type == orc.STDSEG_NO_DUR:
                rc, g, n, z = orc.segtrans_build_gradient(self.ocfg, self.olay, self.lam, self.windows(u), self.labels[u], T, grad=g)
            else:
                rc, g, n, z = orc.seg_build_gradient(self.ocfg, self.olay, self.lam, self.windows(u), self.labels[u], T, grad=g)
            assert rc == 0, rc
            numer.append(n); zx.append(z)
        return g, np.array(numer), np.array(zx)

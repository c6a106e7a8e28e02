"""-m gpu: SURVEY row f3, third step -- the n-state frame model (crf_states = K > 1: nodes/CRF_StdNStateNode.cpp, the
sparse transition layout of ftrmaps/CRF_StdFeatureMap.cpp:280-407, decoders/CRF_LatticeBuilder.h nStateBuildLattice)
through the same engine and C ABI, against the oracle's restatement (tests/test_oracle_nstate.py pins that one by brute
force).  Bars: scores and lattice arcs bit-exact; node values 1e-11; gradient, numerator, Zx 1e-10; Viterbi labels
and float cost identical to the oracle's shortest path."""
import numpy as np
import pytest

import orc
import scrf_amd

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


class NCase:
    def __init__(self, P, K, F, Ts, seed=0, trans_ftrs=True, scale=0.3, scratch_bytes=0):
        rng = np.random.RandomState(seed)
        self.P, self.K, self.L, self.F, self.Ts = P, K, P * K, F, list(Ts)
        self.frames = [rng.random_sample((T, F)).astype(np.float32) for T in Ts]
        kw = dict(model_type=orc.STDFRAME, L=P * K, D=1, F=F, use_trans_ftrs=trans_ftrs, tfs=0, tfe=F - 1, num_states=K)
        self.ocfg = orc.config(**kw); self.olay = orc.Layout(self.ocfg)
        self.gcfg = scrf_amd.make_config(scratch_bytes=scratch_bytes, **kw)
        self.lam = rng.normal(0, scale, self.olay.lambda_len)
        self.labels = []
        for T in Ts:   # sequences the topology allows
            labs = np.zeros(T, dtype=np.uint32)
            c = int(rng.randint(0, P * K))
            for t in range(T):
                labs[t] = c
                r = rng.rand()
                if r >= 0.4:
                    c = int(rng.randint(0, P)) * K if (c + 1) % K == 0 else c + 1
            self.labels.append(labs)
        self.recipes = [scrf_amd.StreamRecipe(F, 0, 0, 0)]

    def engine(self):
        e = scrf_amd.Engine(self.gcfg); e.set_lambda(self.lam); return e

    def batch(self, eng, with_labels=True):
        return eng.batch_from_frames(self.frames, self.labels if with_labels else None, self.recipes, None)


CASES = [dict(P=2, K=2, F=3, Ts=[1, 2, 5]), dict(P=3, K=3, F=4, Ts=[4, 9, 3]), dict(P=16, K=3, F=6, Ts=[20, 7], scale=0.1),
         dict(P=4, K=2, F=3, Ts=[6, 6], trans_ftrs=False), dict(P=2, K=5, F=2, Ts=[12]),
         dict(P=70, K=2, F=3, Ts=[9, 4], scale=0.1)]     # more phones than lane groups in the workgroup


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_scores_node_values_lattice_and_best_path(ci):
    c = NCase(seed=300 + ci, **CASES[ci])
    eng = c.engine(); b = c.batch(eng, with_labels=False)
    assert eng.lambda_len == c.olay.lambda_len
    labs, cost = eng.viterbi_batch(b)
    L, P = c.L, c.P
    for u, T in enumerate(c.Ts):
        So, TDo, TOo, TEo = orc.nstate_scores(c.ocfg, c.olay, c.lam, c.frames[u], T)
        S, M = eng.scores(b, u, T)
        assert np.array_equal(bits(S), bits(So))
        assert np.array_equal(bits(M[:, :L]), bits(TDo)) and np.array_equal(bits(M[:, L:2 * L]), bits(TOo)) and np.array_equal(bits(M[:, 2 * L:]), bits(TEo))
        rc, al, zx = orc.nstate_forward(c.ocfg, So, TDo, TOo, TEo, T)
        rc2, be = orc.nstate_backward(c.ocfg, So, TDo, TOo, TEo, T)
        assert rc == 0 and rc2 == 0
        _, gal, gbe, gzx = eng.forward_backward(b, u, T)
        assert abs(gzx - zx) <= 1e-11 * max(1, abs(zx))
        np.testing.assert_allclose(gal, al, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(gbe, be, rtol=1e-11, atol=1e-11)
        oa, ons, ofin = orc.nstate_lattice_arcs(c.ocfg, So, TDo, TOo, TEo, T)
        ga, gns, gfin = eng.lattice_arcs(b, u)
        assert gns == ons and gfin == ofin and ga.tobytes() == oa.tobytes()
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol) and np.float32(cost[u]) == np.float32(oc) and len(ol) == T
    b.close(); eng.close()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_fb_batch_gradient(ci):
    c = NCase(seed=300 + ci, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og = np.zeros(c.olay.lambda_len); on, oz = [], []
    for u, T in enumerate(c.Ts):
        rc, og, n, z = orc.nstate_build_gradient(c.ocfg, c.olay, c.lam, c.frames[u], c.labels[u], T, grad=og)
        assert rc == 0
        on.append(n); oz.append(z)
    on, oz = np.array(on), np.array(oz)
    tol = 1e-10
    assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - oz).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() <= tol * max(1.0, np.abs(og).max())
    b.close(); eng.close()


def test_chunks_and_refusals():
    kw = dict(P=3, K=2, F=3, Ts=[5, 7, 3, 9, 4, 8])
    out = []
    for sb in (0, 1 << 14):
        c = NCase(seed=9, scratch_bytes=sb, **kw)
        eng = c.engine(); b = c.batch(eng)
        numer, zx = eng.fb_batch(b)
        out.append((numer.copy(), zx.copy(), eng.get_grad().copy()))
        b.close(); eng.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-12, atol=1e-13)
    with pytest.raises(scrf_amd.ScrfError):      # nLabs must be a multiple of the states per label
        scrf_amd.Engine(scrf_amd.make_config(model_type=orc.STDFRAME, L=7, D=1, F=3, num_states=2))
    with pytest.raises(scrf_amd.ScrfError):      # the reference has no n-state node for this model type (tests/test_gpu_segnstate.py has the one it has)
        scrf_amd.Engine(scrf_amd.make_config(model_type=orc.STDSEG_NO_DUR, L=6, D=3, F=8 * 2 + 3, num_states=2))


@pytest.mark.parametrize("prec,tol", [(scrf_amd.PREC_FAST, 1e-9), (scrf_amd.PREC_FAST32, 1e-5), (scrf_amd.PREC_FASTLIN, 1e-9)])
def test_frame_model_recast_as_segmental_with_transition_features(prec, tol):
    """CRFTrain (FAST / FAST32 / FASTLIN, training only) runs the n-state FRAME model as the n-state segmental model
    with maximum duration 1 over the masked dense layout (host/crf_amd.cpp frameAsSegmental): the same function, here with
    transition features (stdtrans), against the oracle's n-state frame gradient; and the recast engine keeps the frame
    node's posterior-mass bounds (scrf_set_frame_mass_check)."""
    c = NCase(P=16, K=3, F=6, Ts=[20, 7, 33], seed=77, trans_ftrs=True, scale=0.1)
    for T, labs in zip(c.Ts, c.labels):   # the segmental node wants an utterance to end in an end state
        labs[T - 1] = (labs[T - 1] // c.K) * c.K + c.K - 1
        for t in range(T - 2, -1, -1):    # and the label sequence to follow the topology back from there
            nxt, cur = int(labs[t + 1]), int(labs[t])
            ok = cur == nxt or (nxt % c.K != 0 and cur == nxt - 1) or (nxt % c.K == 0 and (cur + 1) % c.K == 0)
            if not ok:
                labs[t] = nxt if nxt % c.K == 0 else nxt - 1
    kw = dict(model_type=orc.STDSEG_NO_DUR_NO_SEGTRANSFTR, L=c.L, D=1, F=c.F, use_trans_ftrs=True, tfs=0, tfe=c.F - 1, num_states=c.K)
    eng = scrf_amd.Engine(scrf_amd.make_config(precision=prec, **kw))
    eng.set_frame_mass_check(True)
    assert eng.lambda_len == c.olay.lambda_len
    eng.set_lambda(c.lam)
    b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og = np.zeros(c.olay.lambda_len); on, oz = [], []
    for u, T in enumerate(c.Ts):
        rc, og, n, z = orc.nstate_build_gradient(c.ocfg, c.olay, c.lam, c.frames[u], c.labels[u], T, grad=og)
        assert rc == 0
        on.append(n); oz.append(z)
    assert np.abs(numer - np.array(on)).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - np.array(oz)).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() <= 10 * tol * max(1.0, np.abs(og).max())
    b.close(); eng.close()

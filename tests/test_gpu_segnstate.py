"""-m gpu: SURVEY row f3, last step -- K states per phone on the segmental model (crf_states = K > 1 with
stdseg_no_dur_no_segtransftr: nodes/CRF_StdSegNStateNode_WithoutDurLab_WithoutSegTransFtr.cpp, the compact weight
layout of ftrmaps/CRF_StdFeatureMap.cpp:280-407, nStateBuildLattice of
decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab_WithoutSegTransFtr.h:409-690) through the same engine and C ABI,
against the oracle's restatement (tests/test_oracle_segnstate.py pins that one by brute force).  The engine runs its
dense one-state kernels with the missing transitions' biases at log 0 and shows the caller the compact layout.
Bars: state scores, the topology's transition scores and lattice arcs bit-exact; node values 1e-11; gradient,
numerator, Zx as for the one-state model (1e-9 / 1e-11); Viterbi labels and float cost identical to the shortest path
of the oracle's lattice; optimizer steps bit-exact on the compact vectors."""
import numpy as np
import pytest

import orc
import scrf_amd
from cases import Case

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


# L = phones * K
CASES = [
    dict(L=4, D=3, in_w=2, Ts=[1, 2, 3, 4, 7], num_states=2),
    dict(L=6, D=4, in_w=3, Ts=[3, 4, 5, 12], trans_ctx=1, num_states=3),           # transition features
    dict(L=6, D=1, in_w=4, Ts=[1, 6, 9], trans_ctx=2, num_states=2),                # D = 1 segmental
    dict(L=48, D=25, in_w=39, Ts=[60, 33], num_states=3),                            # config-2 shape, 16 phones x 3
    dict(L=48, D=10, in_w=8, Ts=[40, 25], trans_ctx=1, lam_scale=0.05, num_states=3),
    dict(L=6, D=3, in_w=2, Ts=[9, 14], num_states=2, conform_labels=False),         # labels off the topology
]


def allowed_mask(c):
    L = c.L
    return (c.olay.trans_idx != 0xffffffff).reshape(L, L)      # [previous label, label]


@pytest.fixture(scope="module", params=range(len(CASES)), ids=lambda i: "case%d" % i)
def case(request):
    c = Case(seed=500 + request.param, **CASES[request.param])
    eng = c.engine()
    b = c.batch(eng)
    yield c, eng, b
    b.close(); eng.close()


def test_layout_hooks_show_the_compact_layout(case):
    c, eng, _ = case
    K, L = c.ocfg.num_states, c.L
    P = L // K
    assert eng.lambda_len == c.olay.lambda_len == L * c.olay.num_state_funcs + (P * P + 2 * L - P) * c.olay.num_trans_funcs
    for l in range(L):
        assert eng.state_idx(l) == c.olay.state_idx[l]
        for p in range(L):
            assert eng.trans_idx(p, l) == c.olay.trans_idx[p * L + l]
    # weight-length vectors cross the ABI in that layout
    assert np.array_equal(eng.get_lambda(), c.lam)
    v = np.arange(eng.lambda_len, dtype=np.float64) + 0.5
    eng.set_lambda_acc(v); assert np.array_equal(eng.get_lambda_acc(), v)
    eng.set_grad_sqr_acc(2 * v); assert np.array_equal(eng.get_grad_sqr_acc(), 2 * v)
    eng.set_lambda_acc(np.zeros_like(v)); eng.set_grad_sqr_acc(np.zeros_like(v))


def test_scores_bit_exact_on_the_topology(case):
    c, eng, b = case
    ok = allowed_mask(c).reshape(-1)
    for u, T in enumerate(c.Ts):
        S, M = eng.scores(b, u, T)
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        assert np.array_equal(bits(S), bits(So))
        assert np.array_equal(bits(M[1:, ok]), bits(Mo[1:, ok]))
        assert (M[1:, ~ok] <= -1e29).all()      # log 0 for the transitions the topology lacks


def test_forward_backward(case):
    c, eng, b = case
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        rc, ad, al, apt, zx = orc.seg_forward(c.ocfg, So, Mo, T)
        rc2, be, sd = orc.seg_backward(c.ocfg, So, Mo, T)
        assert rc == 0 and rc2 == 0
        gad, gal, gbe, gzx = eng.forward_backward(b, u, T)
        tol = 1e-11
        assert abs(gzx - zx) <= tol * max(1, abs(zx))
        np.testing.assert_allclose(gad, ad, rtol=tol, atol=tol)
        np.testing.assert_allclose(gal, al, rtol=tol, atol=tol)
        np.testing.assert_allclose(gbe, be, rtol=tol, atol=tol)


@pytest.mark.parametrize("ci", range(len(CASES)))
@pytest.mark.parametrize("prec", [0, 1])
def test_fb_batch_gradient(ci, prec):
    c = Case(seed=500 + ci, precision=prec, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    eng.zero_grad()
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, onumer, ozx = c.oracle_gradient()
    assert g.shape == og.shape
    assert np.abs(numer - onumer).max() <= 1e-11 * max(1, np.abs(onumer).max())
    assert np.abs(zx - ozx).max() <= 1e-11 * np.abs(ozx).max()
    err = np.abs(g - og).max() / np.abs(og).max()
    assert err <= 1e-9, err
    s = eng.batch_sums()
    assert abs(s[0] - onumer.sum()) < 1e-9 * max(1, abs(onumer.sum())) and s[2] == len(c.Ts)
    eng.fb_batch(b)       # accumulates
    np.testing.assert_allclose(eng.get_grad(), 2 * g, rtol=1e-12, atol=1e-12 * np.abs(og).max())
    b.close(); eng.close()


def test_lattice_arcs_bit_exact(case):
    c, eng, b = case
    tot = 0
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        rc, _, _, _, zx = orc.seg_forward(c.ocfg, So, Mo, T)
        for norm in (False, True):
            oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T, norm=norm, alpha_sum=zx)
            ga, gns, gfin = eng.lattice_arcs(b, u, norm=norm)
            assert (gns, gfin) == (ons, ofin)
            if norm:      # the final arcs carry float(Zx): the engine's Zx agrees to 1e-11, not to the bit
                L = c.L
                assert ga[:-L].tobytes() == oa[:-L].tobytes()
                np.testing.assert_allclose(ga[-L:]["w"], oa[-L:]["w"], rtol=1e-6)
            else:
                assert ga.tobytes() == oa.tobytes()
        tot += len(oa)
    assert b.n_arcs == tot


def test_viterbi_matches_shortest_path_on_reference_lattice(case):
    c, eng, b = case
    labs, cost = eng.viterbi_batch(b)
    K, L = c.ocfg.num_states, c.L
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol)
        assert np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()
        seq = [int(x) % L for x in labs[u]]
        assert all(orc.ns_allowed(K, p, q) for p, q in zip(seq, seq[1:]))


def test_minibatch_reduce_and_optimizer_step():
    c = Case(L=6, D=3, in_w=2, Ts=[6, 5, 8], seed=33, num_states=3)
    eng = c.engine(); b = c.batch(eng)
    eng.fb_batch(b)
    g = eng.get_grad()
    s = eng.allreduce_grad(active=True)          # single rank: grad /= 1
    assert s[3] == 1 and np.array_equal(eng.get_grad(), g)
    lam = c.lam.copy(); acc = np.zeros_like(lam); gsa = np.zeros_like(lam); gg = g.copy()
    orc.sgd_step(lam, acc, gsa, gg, 0.1, False)
    eng.sgd_step(0.1, False)
    assert np.array_equal(eng.get_lambda(), lam) and np.array_equal(eng.get_lambda_acc(), acc)
    assert not eng.get_grad().any()
    eng.fb_batch(b)
    g2 = eng.get_grad(); gg = g2.copy()
    og = np.zeros_like(lam)
    for u, T in enumerate(c.Ts):     # the gradient at the stepped weights: the mask survived the step
        rc, og, _, _ = orc.seg_build_gradient(c.ocfg, c.olay, lam, c.windows(u), c.labels[u], T, grad=og)
    assert np.abs(g2 - og).max() <= 1e-9 * np.abs(og).max()
    orc.sgd_step(lam, acc, gsa, gg, 1.0, True, 1e-12)
    eng.sgd_step(1.0, True, 1e-12)
    assert np.array_equal(eng.get_lambda(), lam) and np.array_equal(eng.get_grad_sqr_acc(), gsa)
    assert np.array_equal(eng.get_lambda_acc(), acc)
    # add_grad takes the compact layout too
    eng.zero_grad()
    eng.add_grad(g); eng.add_grad(g)
    assert np.array_equal(eng.get_grad(), 2 * g)
    b.close(); eng.close()


def test_benchmarked_path_serves_the_masked_model():
    """Config-2 shape with 3 states per phone in FAST precision: the batch goes through the fused linear-domain kernels of
    the one-state model (no log-domain redo: a log-0 transition flushes to an exact 0 there), gradient vs the oracle."""
    c = Case(L=48, D=25, in_w=39, Ts=[300, 120, 64], seed=77, precision=1, num_states=3, lam_scale=0.1)
    eng = c.engine(); b = c.batch(eng)
    assert eng.batch_is_fused(b)
    numer, zx = eng.fb_batch(b)
    assert eng.train_stats() == 0
    og, onumer, ozx = c.oracle_gradient()
    assert np.abs(numer - onumer).max() <= 1e-11 * max(1, np.abs(onumer).max())
    assert np.abs(zx - ozx).max() <= 1e-11 * np.abs(ozx).max()
    assert np.abs(eng.get_grad() - og).max() <= 1e-9 * np.abs(og).max()
    b.close(); eng.close()


def test_random_shape_sweep():
    """20 seeded random shapes (phones, states per phone, durations, stream width, lengths incl. shorter than D; with
    and without transition features; both precisions; labels on and off the topology): gradient, numerator, Zx, lattice
    arcs and best path against the oracle."""
    import os
    rng = np.random.RandomState(int(os.environ.get("SCRF_SWEEP_SEED", "2024")))
    for it in range(int(os.environ.get("SCRF_SWEEP_N", "20"))):
        K = int(rng.randint(2, 5)); P = int(rng.randint(1, 9)); D = int(rng.randint(1, 13)); W = int(rng.randint(1, 6))
        Ts = [int(rng.randint(1, 3 * D + 4)) for _ in range(int(rng.randint(1, 5)))]
        kw = dict(L=P * K, D=D, in_w=W, Ts=Ts, num_states=K, seed=9000 + it, precision=int(rng.randint(0, 2)),
                  conform_labels=bool(rng.randint(0, 2)), lam_scale=float(rng.choice([0.05, 0.3, 1.0])))
        if rng.randint(0, 2):
            kw["trans_ctx"] = int(rng.randint(0, 3))
        c = Case(**kw)
        eng = c.engine(); b = c.batch(eng)
        numer, zx = eng.fb_batch(b)
        g = eng.get_grad()
        og, onumer, ozx = c.oracle_gradient()
        assert np.abs(numer - onumer).max() <= 1e-10 * max(1, np.abs(onumer).max()), kw
        assert np.abs(zx - ozx).max() <= 1e-10 * max(1, np.abs(ozx).max()), kw
        assert np.abs(g - og).max() <= 1e-8 * max(1e-300, np.abs(og).max()), kw
        labs, cost = eng.viterbi_batch(b)
        for u, T in enumerate(c.Ts):
            So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
            oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
            ga, gns, gfin = eng.lattice_arcs(b, u)
            assert (gns, gfin) == (ons, ofin) and ga.tobytes() == oa.tobytes(), kw
            ol, oc = orc.best_path(oa, ons, ofin)
            assert np.float32(cost[u]).tobytes() == np.float32(oc).tobytes(), kw
            assert list(labs[u]) == list(ol), kw
        b.close(); eng.close()


def test_refusals():
    kw = dict(L=6, D=3, F=orc.window_width(2, 3, 0, 0, True), num_states=2)
    # the device gradient pointer is a dense-layout buffer: not handed out
    eng = scrf_amd.Engine(scrf_amd.make_config(model_type=orc.STDSEG_NO_DUR_NO_SEGTRANSFTR, **kw))
    with pytest.raises(scrf_amd.ScrfError, match="crf_states > 1"):
        eng.grad_device_ptr()
    eng.close()
    # the topology is held by the transition bias
    with pytest.raises(scrf_amd.ScrfError, match="transition bias"):
        scrf_amd.Engine(scrf_amd.make_config(model_type=orc.STDSEG_NO_DUR_NO_SEGTRANSFTR, use_trans_bias=False, use_trans_ftrs=True, **kw))
    # the reference's other segmental n-state nodes throw "has not been implemented yet" (nodes/CRF_StateNode.cpp:496-507)
    for mt in (orc.STDSEG_NO_DUR, orc.STDSEG):
        with pytest.raises(scrf_amd.ScrfError, match="not been implemented"):
            scrf_amd.Engine(scrf_amd.make_config(model_type=mt, **dict(kw, L=6)))
    with pytest.raises(scrf_amd.ScrfError, match="Invalid state/label combination"):
        scrf_amd.Engine(scrf_amd.make_config(model_type=orc.STDSEG_NO_DUR_NO_SEGTRANSFTR, **dict(kw, L=7)))

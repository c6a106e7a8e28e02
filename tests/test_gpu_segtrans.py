"""-m gpu: SURVEY row f3, first step -- the STDSEG_NO_DUR model (transition features from the segment's own
window: one L x L matrix per window, nodes/CRF_StdSegStateNode_WithoutDurLab.cpp, with the gradbuilder that
passes the previous label, trainers/gradbuilders/CRF_NewGradBuilder_StdSeg.cpp) through the same engine and C
ABI, against the oracle's restatement (tests/test_oracle_segtrans.py pins that one by brute force).
Bars: scores bit-exact (EXACT); node values 1e-11; gradient, numerator, Zx 1e-10 (EXACT) / 1e-9 (FAST)."""
import numpy as np
import pytest

import orc
import scrf_amd
from cases import Case

pytestmark = pytest.mark.gpu
NO_DUR = orc.STDSEG_NO_DUR


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


CASES = [
    dict(L=3, D=3, in_w=2, Ts=[1, 2, 3, 4, 7], trans_share=(0, 18)),              # whole window as transition features
    dict(L=2, D=4, in_w=3, Ts=[3, 4, 5, 12], trans_ctx=1),                        # second stream: boundary context
    dict(L=5, D=2, in_w=4, Ts=[1, 6, 9], trans_share=(4, 19)),                    # a sub-range of the window
    dict(L=7, D=10, in_w=3, Ts=[9, 10, 11, 30], trans_share=(0, 9)),              # T = D-1, D, D+1, 3D
    dict(L=48, D=10, in_w=8, Ts=[40, 25], trans_ctx=1, lam_scale=0.05),           # TIMIT-like label count
    dict(L=4, D=3, in_w=2, Ts=[5, 8]),                                            # bias-only transitions
    dict(L=70, D=3, in_w=2, Ts=[7, 4], trans_share=(0, 3), lam_scale=0.1),        # more labels than lanes; 16-deep load batches + tail
    dict(L=5, D=20, in_w=2, Ts=[50, 19], trans_share=(0, 5), lam_scale=0.2),      # more windows than wavefronts
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_scores_and_node_values(ci):
    c = Case(seed=700 + ci, model_type=NO_DUR, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    assert eng.lambda_len == c.olay.lambda_len
    for u, T in enumerate(c.Ts):
        So, Mo = orc.segtrans_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        S, M = eng.scores(b, u, T)
        assert np.array_equal(bits(S), bits(So)) and np.array_equal(bits(M), bits(Mo))
        rc, ad, al, zx = orc.segtrans_forward(c.ocfg, So, Mo, T)
        rc2, be = orc.segtrans_backward(c.ocfg, So, Mo, T)
        assert rc == 0 and rc2 == 0
        gad, gal, gbe, gzx = eng.forward_backward(b, u, T)
        assert abs(gzx - zx) <= 1e-11 * max(1, abs(zx))
        np.testing.assert_allclose(gad, ad, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(gal, al, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(gbe, be, rtol=1e-11, atol=1e-11)
    b.close(); eng.close()


@pytest.mark.parametrize("prec,tol", [(0, 1e-10), (1, 1e-9)], ids=["exact", "fast"])
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_fb_batch_gradient(ci, prec, tol):
    c = Case(seed=700 + ci, model_type=NO_DUR, precision=prec, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, on, oz = c.oracle_gradient()
    assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - oz).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() <= tol * np.abs(og).max()
    s = eng.batch_sums()
    assert s[2] == len(c.Ts) and abs(s[1] - oz.sum()) <= tol * abs(oz.sum())
    eng.fb_batch(b, want_scalars=False)          # accumulates
    np.testing.assert_allclose(eng.get_grad(), 2 * g, rtol=1e-12, atol=1e-12 * np.abs(g).max())
    assert eng.train_stats() == 0
    b.close(); eng.close()


def test_chunking_sgd_and_errors():
    kw = dict(L=4, D=3, in_w=3, Ts=[6, 9, 4, 12, 7], trans_share=(0, 26), seed=31, model_type=NO_DUR)
    c1 = Case(**kw); c2 = Case(scratch_bytes=1 << 16, **kw)
    e1, e2 = c1.engine(), c2.engine()
    b1, b2 = c1.batch(e1), c2.batch(e2)
    n1, z1 = e1.fb_batch(b1); n2, z2 = e2.fb_batch(b2)
    assert np.array_equal(n1, n2) and np.array_equal(z1, z2)
    np.testing.assert_allclose(e1.get_grad(), e2.get_grad(), rtol=1e-12, atol=1e-13)
    # one SGD step against the oracle's
    og, on, oz = c1.oracle_gradient()
    lam = c1.lam.copy(); acc = np.zeros_like(lam); gsa = np.zeros_like(lam); gg = e1.get_grad()
    orc.sgd_step(lam, acc, gsa, gg, 0.05, False)
    e1.sgd_step(0.05, False)
    assert np.array_equal(e1.get_lambda(), lam)
    # a label out of range is reported by the batch call and leaves the gradient alone
    labs = [l.copy() for l in c1.labels]; labs[2][-1] = c1.L * c1.D + 1
    bad = e2.batch_from_frames(c1.frames, labs, c1.recipes)
    g0 = e2.get_grad()
    with pytest.raises(scrf_amd.ScrfError) as ei:
        e2.fb_batch(bad, want_scalars=False)
    assert ei.value.code == 5 and np.array_equal(e2.get_grad(), g0)
    for x in (b1, b2, bad): x.close()
    e1.close(); e2.close()


def test_bias_only_transitions_equal_the_timit_demo_model():
    """without transition features the per-window matrices are all the same bias matrix: the STDSEG_NO_DUR
    engine path and the STDSEG_NO_DUR_NO_SEGTRANSFTR path must agree (different kernels, same model)"""
    kw = dict(L=6, D=4, in_w=3, Ts=[9, 14, 3], seed=5)
    c1 = Case(model_type=NO_DUR, **kw); c2 = Case(**kw)
    e1, e2 = c1.engine(), c2.engine()
    b1, b2 = c1.batch(e1), c2.batch(e2)
    n1, z1 = e1.fb_batch(b1); n2, z2 = e2.fb_batch(b2)
    np.testing.assert_allclose(n1, n2, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(z1, z2, rtol=1e-12)
    g1, g2 = e1.get_grad(), e2.get_grad()
    assert np.abs(g1 - g2).max() <= 1e-10 * np.abs(g2).max()
    for x in (b1, b2): x.close()
    e1.close(); e2.close()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_lattice_arcs_and_best_path(ci):
    """decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab.h: arcs byte for byte in AddArc order (also with the
    normaliser on the final arcs), and the batched best path == ShortestPath on the oracle's lattice."""
    c = Case(seed=700 + ci, model_type=NO_DUR, **CASES[ci])
    eng = c.engine(); b = c.batch(eng, with_labels=False)
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.segtrans_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.segtrans_lattice_arcs(c.ocfg, So, Mo, T)
        ga, gns, gfin = eng.lattice_arcs(b, u)
        assert (gns, gfin) == (ons, ofin) and ga.tobytes() == oa.tobytes()
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol) and np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()
        if u == 0:
            rc, ad, al, zx = orc.segtrans_forward(c.ocfg, So, Mo, T)
            oan, _, _ = orc.segtrans_lattice_arcs(c.ocfg, So, Mo, T, norm=True, alpha_sum=zx)
            gan, _, _ = eng.lattice_arcs(b, u, norm=True)
            assert gan[:-c.L].tobytes() == oan[:-c.L].tobytes()
            np.testing.assert_allclose(gan["w"][-c.L:], oan["w"][-c.L:], rtol=1e-6)
    b.close(); eng.close()

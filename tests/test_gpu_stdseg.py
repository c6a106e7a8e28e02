"""-m gpu: SURVEY row f3, second step -- the STDSEG model (labels carry the duration: own state weights per
(phone, duration), transitions over full labels from the segment's own window, nodes/CRF_StdSegStateNode.cpp, with
trainers/gradbuilders/CRF_NewGradBuilder_StdSeg.cpp) through the same engine and C ABI, training side and node values,
against the oracle's restatement (tests/test_oracle_stdseg.py pins that one by brute force), plus its lattice
(decoders/CRF_LatticeBuilder_StdSeg.h) and best path.  Bars: scores and lattice arcs bit-exact; node values 1e-11;
gradient, numerator, Zx 1e-10; Viterbi labels and float cost identical to the oracle's shortest path."""
import numpy as np
import pytest

import orc
import scrf_amd
from cases import Case

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint64)


CASES = [
    dict(L=3, D=3, in_w=2, Ts=[1, 2, 3, 4, 7], trans_share=(0, 1)),          # transition features: two window columns
    dict(L=2, D=4, in_w=3, Ts=[3, 4, 5, 12], trans_ctx=1),                    # second stream: boundary context
    dict(L=5, D=2, in_w=4, Ts=[1, 6, 9], trans_share=(4, 9)),
    dict(L=4, D=5, in_w=3, Ts=[4, 5, 6, 15]),                                 # T = D-1, D, D+1, 3D; bias-only transitions
    dict(L=12, D=4, in_w=4, Ts=[20, 9], lam_scale=0.1),                       # 48 full labels
    dict(L=70, D=2, in_w=2, Ts=[5, 3], lam_scale=0.1),                        # more phones than lanes (two lane rounds), 33+ previous labels per walk
    dict(L=9, D=12, in_w=2, Ts=[30], lam_scale=0.1),                          # 108 previous labels: the 32-deep load batches and their tail
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_scores_and_node_values(ci):
    c = Case(seed=900 + ci, model_type=orc.STDSEG, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    assert eng.lambda_len == c.olay.lambda_len
    for u, T in enumerate(c.Ts):
        So, Mo = orc.stdseg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        S, M = eng.scores(b, u, T)
        assert np.array_equal(bits(S), bits(So)) and np.array_equal(bits(M), bits(Mo))
        rc, al, zx = orc.stdseg_forward(c.ocfg, So, Mo, T)
        rc2, be = orc.stdseg_backward(c.ocfg, So, Mo, T)
        assert rc == 0 and rc2 == 0
        gal, _, gbe, gzx = eng.forward_backward(b, u, T)
        assert abs(gzx - zx) <= 1e-11 * max(1, abs(zx))
        np.testing.assert_allclose(gal, al, rtol=1e-11, atol=1e-11)
        # beta is defined on the labels a node can carry: rows of the node; all of them here
        np.testing.assert_allclose(gbe, be, rtol=1e-11, atol=1e-11)
    b.close(); eng.close()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_fb_batch_gradient(ci):
    c = Case(seed=900 + ci, model_type=orc.STDSEG, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, on, oz = c.oracle_gradient()
    tol = 1e-10
    assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - oz).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() <= tol * max(1.0, np.abs(og).max())
    # a second batch accumulates; sums follow
    numer2, zx2 = eng.fb_batch(b)
    np.testing.assert_allclose(eng.get_grad(), 2 * g, rtol=1e-12, atol=1e-13)
    s = eng.batch_sums()
    assert abs(s[0] - 2 * on.sum()) <= 1e-9 * max(1, abs(on.sum())) and s[2] == 2 * len(c.Ts)
    b.close(); eng.close()


LIN_CASES = [
    dict(L=4, D=5, in_w=3, Ts=[4, 5, 6, 15, 1, 2]),          # T = 1, 2, D-1, D, D+1, 3D
    dict(L=12, D=4, in_w=4, Ts=[20, 9], lam_scale=0.1),      # 48 full labels
    dict(L=48, D=10, in_w=5, Ts=[40, 9, 10, 11], lam_scale=0.05),   # the TIMIT label space: 480 full labels (one thread each)
    dict(L=3, D=3, in_w=2, Ts=[7, 3, 5, 8, 2, 9, 4]),
    dict(L=9, D=12, in_w=2, Ts=[30, 13], lam_scale=0.1),
]


@pytest.mark.parametrize("prec,tol", [(scrf_amd.PREC_FAST, 1e-9), (scrf_amd.PREC_FASTLIN, 1e-9), (scrf_amd.PREC_FAST32, 1e-5)])
@pytest.mark.parametrize("ci", range(len(LIN_CASES)))
def test_bias_only_transitions_linear_domain_path(ci, prec, tol, monkeypatch):
    """FAST precisions with bias-only transitions (`stdstate`): scrf_stdseg_lin.hip -- one exp(M) table, mantissa
    recursion as a matrix-vector product per node, duration-major node arrays through the dense MFMA contractions,
    transition counts as E o (A^T B).  Against the oracle (the reference's log-domain order), and against the
    reference-order kernels of the same engine (SCRF_STDSEG_LIN=0)."""
    c = Case(seed=940 + ci, model_type=orc.STDSEG, precision=prec, **LIN_CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, on, oz = c.oracle_gradient()
    assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - oz).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() <= 10 * tol * max(1.0, np.abs(og).max())
    numer2, zx2 = eng.fb_batch(b)   # accumulates
    np.testing.assert_allclose(eng.get_grad(), 2 * g, rtol=1e-12, atol=1e-13)
    assert eng.batch_sums()[2] == 2 * len(c.Ts)
    # which kernels ran
    eng.enable_timing(True); eng.zero_grad(); eng.fb_batch(b)
    names = [k[0] for k in eng.kernel_timing()]
    eng.enable_timing(False)
    assert "k_sl_fb" in names and "k_stdseg_fb" not in names
    b.close(); eng.close()
    if prec == scrf_amd.PREC_FAST:
        monkeypatch.setenv("SCRF_STDSEG_LIN", "0")
        eng = c.engine(); b = c.batch(eng)
        n0, z0 = eng.fb_batch(b)
        np.testing.assert_allclose(numer, n0, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(zx, z0, rtol=1e-11)
        assert np.abs(eng.get_grad() - g).max() <= 1e-9 * max(1.0, np.abs(g).max())
        b.close(); eng.close()


def test_linear_domain_path_chunks_errors_and_extreme_weights():
    kw = dict(L=3, D=3, in_w=2, Ts=[5, 7, 3, 9, 4, 8])
    out = []
    for sb in (0, 1 << 16):     # one chunk / a few utterances per chunk
        c = Case(seed=78, model_type=orc.STDSEG, precision=1, scratch_bytes=sb, **kw)
        eng = c.engine(); b = c.batch(eng)
        numer, zx = eng.fb_batch(b)
        out.append((numer.copy(), zx.copy(), eng.get_grad().copy()))
        b.close(); eng.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-12, atol=1e-13)
    # a label outside nLabs: the batch fails and contributes nothing
    c = Case(seed=5, model_type=orc.STDSEG, precision=1, L=3, D=3, in_w=2, Ts=[6, 5])
    eng = c.engine()
    bad = [l.copy() for l in c.labels]
    bad[1][-1] = 3 * 3 + 2
    b = eng.batch_from_frames(c.frames, bad, c.recipes, None)
    with pytest.raises(scrf_amd.ScrfError) as ei:
        eng.fb_batch(b)
    assert "label" in str(ei.value).lower() and np.all(eng.get_grad() == 0.0)
    b.close(); eng.close()
    # weights two orders of magnitude larger (scores in the hundreds): the per-node log-scales carry the range
    c = Case(seed=6, model_type=orc.STDSEG, precision=1, L=4, D=4, in_w=3, Ts=[25, 8], lam_scale=30.0)
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    og, on, oz = c.oracle_gradient()
    assert np.abs(zx - oz).max() <= 1e-10 * np.abs(oz).max()
    assert np.abs(eng.get_grad() - og).max() <= 1e-8 * max(1.0, np.abs(og).max())
    b.close(); eng.close()


def test_chunked_batches_equal_one_chunk():
    kw = dict(L=3, D=3, in_w=2, Ts=[5, 7, 3, 9, 4, 8], trans_share=(0, 1))
    c1 = Case(seed=77, model_type=orc.STDSEG, **kw)
    c2 = Case(seed=77, model_type=orc.STDSEG, scratch_bytes=1 << 16, **kw)     # a few utterances per chunk
    out = []
    for c in (c1, c2):
        eng = c.engine(); b = c.batch(eng)
        numer, zx = eng.fb_batch(b)
        out.append((numer.copy(), zx.copy(), eng.get_grad().copy()))
        b.close(); eng.close()
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-12, atol=1e-13)


def test_errors_and_refusals():
    c = Case(seed=5, model_type=orc.STDSEG, L=3, D=3, in_w=2, Ts=[6, 5])
    eng = c.engine()
    # a label outside nLabs
    bad = [l.copy() for l in c.labels]
    bad[1][-1] = 3 * 3 + 2
    b = eng.batch_from_frames(c.frames, bad, c.recipes, None)
    with pytest.raises(scrf_amd.ScrfError) as ei:
        eng.fb_batch(b)
    assert "label" in str(ei.value).lower()
    assert np.all(eng.get_grad() == 0.0)          # a failed batch contributes nothing
    b.close()
    eng.close()
    # nLabs must be a multiple of the maximum duration (nodes/CRF_StdSegStateNode.cpp:34-40)
    with pytest.raises(scrf_amd.ScrfError):
        scrf_amd.Engine(scrf_amd.make_config(model_type=orc.STDSEG, L=10, D=3, F=19))


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_lattice_arcs_and_best_path(ci):
    c = Case(seed=900 + ci, model_type=orc.STDSEG, **CASES[ci])
    eng = c.engine(); b = c.batch(eng, with_labels=False)
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.stdseg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.stdseg_lattice_arcs(c.ocfg, So, Mo, T)
        ga, gns, gfin = eng.lattice_arcs(b, u)
        assert gns == ons and gfin == ofin and ga.tobytes() == oa.tobytes()
        rc, al, zx = orc.stdseg_forward(c.ocfg, So, Mo, T)
        oa2, _, _ = orc.stdseg_lattice_arcs(c.ocfg, So, Mo, T, norm=True, alpha_sum=zx)
        ga2, _, _ = eng.lattice_arcs(b, u, norm=True)
        nlast = c.L * min(T, c.D)
        assert ga2[:-nlast].tobytes() == oa2[:-nlast].tobytes()
        np.testing.assert_allclose(ga2["w"][-nlast:], oa2["w"][-nlast:], rtol=1e-6)
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol) and np.float32(cost[u]) == np.float32(oc)
        # the labels tile the utterance
        assert sum(int(x) // c.L + 1 for x in labs[u]) == T
    b.close(); eng.close()

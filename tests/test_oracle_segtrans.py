"""Oracle self-consistency for SURVEY row f3, first step: the STDSEG_NO_DUR model
(nodes/CRF_StdSegStateNode_WithoutDurLab.cpp -- transition features taken from the segment's own window,
one L x L matrix per duration -- with trainers/gradbuilders/CRF_NewGradBuilder_StdSeg.cpp, which passes the
PREVIOUS label).  PARITY UNPINNED against the reference binary (oracle/scrf_oracle.h); cross-checked by
brute-force enumeration of all labelled segmentations, central finite differences, and the node's own
posterior-mass self-checks (:530-545)."""
import numpy as np
import pytest

import orc
from scrf_amd import synth


def _case(L, D, T, in_w, seed=0, scale=0.3, share=True):
    """one segment-recipe stream; state AND transition features are the same window vector (share) or the
    transition features a sub-range of it"""
    rng = np.random.RandomState(seed)
    frames = rng.random_sample((T, in_w)).astype(np.float32)
    F = orc.window_width(in_w, D, 0, 0, True)
    X = orc.windows(frames, D)
    cfg = orc.config(model_type=orc.STDSEG_NO_DUR, L=L, D=D, F=F, use_trans_ftrs=True, tfs=0 if share else in_w, tfe=F - 1 if share else 3 * in_w - 1)
    lay = orc.Layout(cfg)
    lam = rng.normal(0, scale, lay.lambda_len)
    labels = synth.group_labels(synth.frame_labels(rng, T, L, D), D, L)
    return cfg, lay, lam, X, labels


@pytest.mark.parametrize("L,D,T", [(2, 2, 4), (3, 3, 6), (2, 3, 5), (3, 2, 5), (2, 1, 4), (3, 3, 1), (3, 3, 2), (2, 4, 7)])
def test_forward_backward_vs_enumeration(L, D, T):
    cfg, lay, lam, X, _ = _case(L, D, T, 2, seed=L * 100 + D * 10 + T)
    S, M2 = orc.segtrans_scores(cfg, lay, lam, X, T)
    bf = orc.brute_force_segtrans(S, M2, T, L, D)
    rc, ad, al, zx = orc.segtrans_forward(cfg, S, M2, T)
    assert rc == 0 and abs(zx - bf["Zx"]) < 1e-12 * max(1, abs(zx))
    rc, g, xi, zx2 = orc.segtrans_posteriors(cfg, S, M2, T)
    assert rc == 0 and zx2 == zx
    np.testing.assert_allclose(g, bf["gamma"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(xi, bf["xi"], rtol=0, atol=1e-12)
    # beta: Zx is also the log-sum over the utterance-initial segments of S + beta
    rc, beta = orc.segtrans_backward(cfg, S, M2, T)
    assert rc == 0
    terms = [S[orc.seg_base(d - 1, D) + d - 1, l] + beta[d - 1, l] for d in range(1, min(D, T) + 1) for l in range(L)]
    mx = max(terms)
    assert abs(mx + np.log(sum(np.exp(x - mx) for x in terms)) - zx) < 1e-12 * max(1, abs(zx))
    # the node's self checks: state mass per end frame <= 1, == transition mass into it (first node: 1)
    for t in range(T):
        b = orc.seg_base(t, D); nd = min(t + 1, D); npv = min(t, D)
        sm = g[b:b + nd].sum()
        assert sm <= 1 + 1e-9
        tm = xi[b:b + npv].sum()
        assert tm <= 1 + 1e-9 and tm <= sm + 1e-9        # the initial segment carries state mass but no transition


@pytest.mark.parametrize("L,D,T,share", [(3, 3, 7, True), (3, 2, 6, False), (4, 4, 9, True)])
def test_gradient_is_derivative_of_loglik(L, D, T, share):
    cfg, lay, lam, X, labels = _case(L, D, T, 2, seed=5 + T, share=share)
    rc, grad, numer, zx = orc.segtrans_build_gradient(cfg, lay, lam, X, labels, T)
    assert rc == 0
    rng = np.random.RandomState(1)
    idx = rng.choice(lay.lambda_len, size=min(40, lay.lambda_len), replace=False)
    eps = 1e-6
    for i in idx:
        lp = lam.copy(); lp[i] += eps
        lm = lam.copy(); lm[i] -= eps
        _, _, n1, z1 = orc.segtrans_build_gradient(cfg, lay, lp, X, labels, T)
        _, _, n0, z0 = orc.segtrans_build_gradient(cfg, lay, lm, X, labels, T)
        fd = ((n1 - z1) - (n0 - z0)) / (2 * eps)
        assert abs(fd - grad[i]) < 2e-6 * max(1.0, abs(grad[i])), (i, fd, grad[i])


def test_numerator_is_score_of_reference_path_and_gradient_accumulates():
    L, D, T = 3, 3, 9
    cfg, lay, lam, X, labels = _case(L, D, T, 2, seed=11)
    S, M2 = orc.segtrans_scores(cfg, lay, lam, X, T)
    _, g1, numer, zx = orc.segtrans_build_gradient(cfg, lay, lam, X, labels, T)
    tot, prev = 0.0, None
    for t in range(T):
        if labels[t] == orc.LAB_BAD:
            continue
        l, d = int(labels[t]) % L, int(labels[t]) // L + 1
        row = orc.seg_base(t, D) + d - 1
        tot += S[row, l]
        if prev is not None:
            tot += M2[row, prev * L + l]
        prev = l
    assert abs(tot - numer) < 1e-12 * max(1, abs(tot)) and numer <= zx + 1e-9
    pre = np.full(lay.lambda_len, 2.5)
    _, g2, _, _ = orc.segtrans_build_gradient(cfg, lay, lam, X, labels, T, grad=pre.copy())
    np.testing.assert_allclose(g2, g1 + 2.5, rtol=1e-13, atol=1e-13)


def test_without_transition_features_it_is_the_no_segtransftr_model():
    """bias-only transitions do not depend on the window: STDSEG_NO_DUR then equals
    STDSEG_NO_DUR_NO_SEGTRANSFTR (same Zx, same gradient)"""
    L, D, T = 3, 3, 8
    rng = np.random.RandomState(4)
    frames = rng.random_sample((T, 2)).astype(np.float32)
    X = orc.windows(frames, D)
    F = X.shape[1]
    c1 = orc.config(model_type=orc.STDSEG_NO_DUR, L=L, D=D, F=F)
    c2 = orc.config(model_type=orc.STDSEG_NO_DUR_NO_SEGTRANSFTR, L=L, D=D, F=F)
    l1, l2 = orc.Layout(c1), orc.Layout(c2)
    assert l1.lambda_len == l2.lambda_len
    lam = rng.normal(0, 0.3, l1.lambda_len)
    labels = synth.group_labels(synth.frame_labels(rng, T, L, D), D, L)
    _, g1, n1, z1 = orc.segtrans_build_gradient(c1, l1, lam, X, labels, T)
    _, g2, n2, z2 = orc.seg_build_gradient(c2, l2, lam, X, labels, T)
    assert abs(z1 - z2) < 1e-12 * abs(z2) and abs(n1 - n2) < 1e-12 * max(1, abs(n2))
    np.testing.assert_allclose(g1, g2, rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("L,D,T", [(2, 2, 4), (3, 3, 6), (2, 3, 5), (3, 2, 1), (2, 4, 7)])
def test_lattice_structure_and_best_path(L, D, T):
    """decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab.h: one state per (node, label); the shortest path's
    cost is -(best path score) of the enumeration and its labels spell the best segmentation; with norm the
    final arcs carry Zx."""
    cfg, lay, lam, X, _ = _case(L, D, T, 2, seed=40 + L * 10 + T)
    S, M2 = orc.segtrans_scores(cfg, lay, lam, X, T)
    arcs, ns, fin = orc.segtrans_lattice_arcs(cfg, S, M2, T)
    assert ns == 1 + T * L + 1 and fin == ns - 1
    assert np.all(arcs["dst"] > arcs["src"])                      # state ids are a topological order
    bf = orc.brute_force_segtrans(S, M2, T, L, D)
    labs, cost = orc.best_path(arcs, ns, fin)
    sc, segs = bf["best"]
    assert [int(x) for x in labs] == [l + L * (d - 1) for (_, d, l) in segs]
    assert abs(-cost - sc) < 1e-5 * max(1, abs(sc))
    rc, ad, al, zx = orc.segtrans_forward(cfg, S, M2, T)
    arcs_n, _, _ = orc.segtrans_lattice_arcs(cfg, S, M2, T, norm=True, alpha_sum=zx)
    assert np.array_equal(arcs_n[:-L], arcs[:-L]) and np.all(arcs_n["w"][-L:] == np.float32(zx))

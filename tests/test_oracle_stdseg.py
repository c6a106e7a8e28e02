"""Oracle self-consistency for SURVEY row f3, second step: the STDSEG model (nodes/CRF_StdSegStateNode.cpp --
labels carry the duration, clab = (dur-1)*nActualLabs + phone, with their own state weights and a transition matrix
over FULL labels taken from the segment's own window -- with trainers/gradbuilders/CRF_NewGradBuilder_StdSeg.cpp).
PARITY UNPINNED against the reference binary (oracle/scrf_oracle.h); cross-checked by brute-force enumeration of all
labelled segmentations, central finite differences and the node's posterior-mass self-checks (:402-421)."""
import numpy as np
import pytest

import orc
from scrf_amd import synth


def _case(L, D, T, in_w, seed=0, scale=0.3, trans_ftrs=True):
    rng = np.random.RandomState(seed)
    frames = rng.random_sample((T, in_w)).astype(np.float32)
    F = orc.window_width(in_w, D, 0, 0, True)
    X = orc.windows(frames, D)
    cfg = orc.config(model_type=orc.STDSEG, L=L * D, D=D, F=F, use_trans_ftrs=trans_ftrs, tfs=0, tfe=in_w - 1)
    lay = orc.Layout(cfg)
    lam = rng.normal(0, scale, lay.lambda_len)
    labels = synth.group_labels(synth.frame_labels(rng, T, L, D), D, L)
    return cfg, lay, lam, X, labels


@pytest.mark.parametrize("L,D,T", [(2, 2, 4), (3, 3, 6), (2, 3, 5), (3, 2, 5), (2, 1, 4), (3, 3, 1), (3, 3, 2), (2, 4, 7)])
def test_forward_backward_vs_enumeration(L, D, T):
    cfg, lay, lam, X, _ = _case(L, D, T, 2, seed=L * 100 + D * 10 + T)
    assert lay.lambda_len == L * D * ((X.shape[1] + 1) + L * D * 3)     # full labels: own state block each, NL x NL transition blocks
    S, MX = orc.stdseg_scores(cfg, lay, lam, X, T)
    bf = orc.brute_force_stdseg(S, MX, T, L, D)
    rc, al, zx = orc.stdseg_forward(cfg, S, MX, T)
    assert rc == 0 and abs(zx - bf["Zx"]) < 1e-12 * max(1, abs(zx))
    rc, beta = orc.stdseg_backward(cfg, S, MX, T)
    assert rc == 0
    # gamma = exp(alpha + beta - Zx) per full label; xi = exp(alpha_prev + MX + S + beta - Zx)
    g = np.exp(al + beta - zx)
    np.testing.assert_allclose(g, bf["gamma"], rtol=0, atol=1e-12)
    for t in range(T):
        b = orc.seg_base(t, D); nd = min(t + 1, D); npv = min(t, D)
        for d in range(1, npv + 1):
            pb = orc.seg_base(t - d, D); pnd = min(t - d + 1, D)
            pa = al[pb:pb + pnd].reshape(-1)
            xi = np.exp(pa[:, None] + MX[b + d - 1, :pnd * L, :] + S[b + d - 1][None, :] + beta[b + d - 1][None, :] - zx)
            np.testing.assert_allclose(xi, bf["xi"][b + d - 1, :pnd * L, :], rtol=0, atol=1e-12)
        assert g[b:b + nd].sum() <= 1 + 1e-9
    # Zx is also the log-sum over the utterance-initial segments of alpha(=S) + beta
    terms = [S[orc.seg_base(d - 1, D) + d - 1, l] + beta[orc.seg_base(d - 1, D) + d - 1, l] for d in range(1, min(D, T) + 1) for l in range(L)]
    mx = max(terms)
    assert abs(mx + np.log(sum(np.exp(x - mx) for x in terms)) - zx) < 1e-12 * max(1, abs(zx))


@pytest.mark.parametrize("L,D,T,tf", [(3, 3, 7, True), (3, 2, 6, False), (2, 4, 9, True)])
def test_gradient_is_derivative_of_loglik(L, D, T, tf):
    cfg, lay, lam, X, labels = _case(L, D, T, 2, seed=5 + T, trans_ftrs=tf)
    rc, grad, numer, zx = orc.stdseg_build_gradient(cfg, lay, lam, X, labels, T)
    assert rc == 0
    rng = np.random.RandomState(1)
    idx = rng.choice(lay.lambda_len, size=min(60, lay.lambda_len), replace=False)
    eps = 1e-6
    for i in idx:
        lp = lam.copy(); lp[i] += eps
        lm = lam.copy(); lm[i] -= eps
        _, _, n1, z1 = orc.stdseg_build_gradient(cfg, lay, lp, X, labels, T)
        _, _, n0, z0 = orc.stdseg_build_gradient(cfg, lay, lm, X, labels, T)
        fd = ((n1 - z1) - (n0 - z0)) / (2 * eps)
        assert abs(fd - grad[i]) < 2e-6 * max(1.0, abs(grad[i])), (i, fd, grad[i])


def test_numerator_is_score_of_reference_path():
    L, D, T = 3, 3, 9
    cfg, lay, lam, X, labels = _case(L, D, T, 2, seed=11)
    S, MX = orc.stdseg_scores(cfg, lay, lam, X, T)
    _, g1, numer, zx = orc.stdseg_build_gradient(cfg, lay, lam, X, labels, T)
    tot, prev = 0.0, None
    for t in range(T):
        if labels[t] == orc.LAB_BAD:
            continue
        l, d = int(labels[t]) % L, int(labels[t]) // L + 1
        row = orc.seg_base(t, D) + d - 1
        tot += S[row, l]
        if prev is not None:
            tot += MX[row, prev, l]
        prev = int(labels[t])
    assert abs(tot - numer) < 1e-11 * max(1, abs(tot))
    assert numer <= zx + 1e-9
    # a second utterance accumulates into the same gradient vector
    _, g2, _, _ = orc.stdseg_build_gradient(cfg, lay, lam, X, labels, T, grad=g1.copy())
    np.testing.assert_allclose(g2, 2 * g1, rtol=1e-12, atol=1e-14)


def test_duration_specific_state_weights_matter_and_collapse():
    """with the state block of every duration set to the same weights and transitions that ignore the previous
    duration, STDSEG equals STDSEG_NO_DUR on the same inputs (the models differ only in what the labels distinguish)"""
    L, D, T, in_w = 3, 3, 7, 2
    cfg, lay, lam, X, _ = _case(L, D, T, in_w, seed=3)
    F = X.shape[1]
    cfg2 = orc.config(model_type=orc.STDSEG_NO_DUR, L=L, D=D, F=F, use_trans_ftrs=True, tfs=0, tfe=in_w - 1)
    lay2 = orc.Layout(cfg2)
    rng = np.random.RandomState(9)
    lam2 = rng.normal(0, 0.3, lay2.lambda_len)
    NL = L * D
    lamf = np.zeros(lay.lambda_len)
    ns, nt = F + 1, in_w + 1
    for c in range(NL):
        cb = c * (ns + NL * nt)          # block of full label c: state funcs, then NL transition blocks (CRF_StdFeatureMap.cpp:280-320)
        c2 = (c % L) * (ns + L * nt)
        lamf[cb:cb + ns] = lam2[c2:c2 + ns]
        for p in range(NL):
            lamf[cb + ns + p * nt: cb + ns + (p + 1) * nt] = lam2[c2 + ns + (p % L) * nt: c2 + ns + (p % L + 1) * nt]
    S, MX = orc.stdseg_scores(cfg, lay, lamf, X, T)
    S2, M2 = orc.segtrans_scores(cfg2, lay2, lam2, X, T)
    _, _, zx = orc.stdseg_forward(cfg, S, MX, T)
    _, _, _, zx2 = orc.segtrans_forward(cfg2, S2, M2, T)
    assert abs(zx - zx2) < 1e-12 * max(1, abs(zx))


@pytest.mark.parametrize("L,D,T", [(2, 2, 4), (3, 3, 6), (2, 3, 5), (3, 3, 1), (2, 4, 7)])
def test_lattice_paths_are_the_labelled_segmentations(L, D, T):
    """decoders/CRF_LatticeBuilder_StdSeg.h: the lattice's paths are exactly the labelled segmentations, each with
    (float sums of) the negated model score; its best path is the enumeration's best segmentation."""
    cfg, lay, lam, X, _ = _case(L, D, T, 2, seed=40 + L * 100 + D * 10 + T)
    S, MX = orc.stdseg_scores(cfg, lay, lam, X, T)
    bf = orc.brute_force_stdseg(S, MX, T, L, D)
    arcs, ns, fin = orc.stdseg_lattice_arcs(cfg, S, MX, T)
    assert ns == 2 + L * orc.num_segs(T, D) and fin == ns - 1
    # arcs leave states in ascending target order and every target is above its source (topological)
    assert np.all(arcs["dst"] > arcs["src"])
    # path count by dynamic programming over the states == number of enumerated segmentations
    cnt = np.zeros(ns); cnt[0] = 1
    for a in arcs:
        cnt[a["dst"]] += cnt[a["src"]]
    assert cnt[fin] == bf["n_paths"]
    ol, cost = orc.best_path(arcs, ns, fin)
    best_sc, best_segs = bf["best"]
    assert abs(-cost - best_sc) < 1e-4 * max(1.0, abs(best_sc))
    scores = sorted(p[0] for p in bf["paths"])
    if len(scores) == 1 or scores[-1] - scores[-2] > 1e-4:
        assert list(ol) == [(d - 1) * L + l for (_, d, l) in best_segs]

"""Oracle self-consistency for SURVEY row f3, last step: K states per phone on the segmental model
(nodes/CRF_StdSegNStateNode_WithoutDurLab_WithoutSegTransFtr.cpp: the one-state node with every transition loop cut
down to the topology -- itself, the state before, or any phone's end state into a start state; the compact weight
layout of ftrmaps/CRF_StdFeatureMap.cpp:280-407; nStateBuildLattice of
decoders/CRF_LatticeBuilder_StdSeg_WithoutDurLab_WithoutSegTransFtr.h:409-690).  PARITY UNPINNED against the reference
binary; cross-checked by brute-force enumeration of the labelled segmentations the topology allows, central finite
differences, and by the dense one-state oracle run with the missing transitions' biases at log 0."""
import numpy as np
import pytest

import orc

MODEL = orc.STDSEG_NO_DUR_NO_SEGTRANSFTR


def allowed_sequence(rng, P, K, n):
    """n segment labels the topology allows."""
    L = P * K
    out = []
    c = int(rng.randint(0, L))
    for _ in range(n):
        out.append(c)
        r = rng.rand()
        if r < 0.3:
            pass
        elif (c + 1) % K == 0:
            c = int(rng.randint(0, P)) * K
        else:
            c += 1
    return out


def frame_labels(rng, P, K, T, D, conform=True):
    """Segment-end labelling (label + L*(dur-1) on a segment's last frame, CRF_LAB_BAD elsewhere)."""
    L = P * K
    durs = []
    left = T
    while left > 0:
        d = int(rng.randint(1, min(D, left) + 1))
        durs.append(d)
        left -= d
    labs = allowed_sequence(rng, P, K, len(durs)) if conform else [int(x) for x in rng.randint(0, L, len(durs))]
    out = np.full(T, 0xffffffff, dtype=np.uint32)
    t = -1
    for d, l in zip(durs, labs):
        t += d
        out[t] = l + L * (d - 1)
    return out


def case(P, K, D, T, F=3, seed=0, trans_ftrs=False, scale=0.4, conform=True):
    rng = np.random.RandomState(seed)
    L = P * K
    X = rng.random_sample((orc.num_segs(T, D), F)).astype(np.float32)
    cfg = orc.config(model_type=MODEL, L=L, D=D, F=F, use_trans_ftrs=trans_ftrs, tfs=0, tfe=F - 1, num_states=K)
    lay = orc.Layout(cfg)
    lam = rng.normal(0, scale, lay.lambda_len)
    labs = frame_labels(rng, P, K, T, D, conform)
    return cfg, lay, lam, X, labs


def dense_twin(cfg, lay, lam):
    """The one-state model over the same labels whose weights equal the compact ones and whose missing transitions
    carry a bias of -1e30."""
    L, K = cfg.num_labs, cfg.num_states
    dcfg = orc.config(model_type=MODEL, L=L, D=cfg.lab_max_dur, F=cfg.num_feas, use_trans_ftrs=bool(cfg.use_trans_ftrs),
                      tfs=cfg.trans_fidx_start, tfe=cfg.trans_fidx_end)
    dlay = orc.Layout(dcfg)
    nsf, ntf = lay.num_state_funcs, lay.num_trans_funcs
    dl = np.zeros(dlay.lambda_len)
    s2d = np.zeros(lay.lambda_len, dtype=np.int64)
    for c in range(L):
        s2d[lay.state_idx[c]:lay.state_idx[c] + nsf] = dlay.state_idx[c] + np.arange(nsf)
        for p in range(L):
            ti = lay.trans_idx[p * L + c]
            if ti == 0xffffffff:
                dl[dlay.trans_idx[p * L + c] + ntf - 1] = -1e30
            else:
                s2d[ti:ti + ntf] = dlay.trans_idx[p * L + c] + np.arange(ntf)
    dl[s2d] = lam
    return dcfg, dlay, dl, s2d


@pytest.mark.parametrize("P,K,D,T", [(2, 2, 2, 4), (2, 2, 3, 5), (3, 2, 2, 4), (2, 3, 2, 5), (2, 2, 3, 1), (2, 2, 3, 2)])
def test_forward_backward_vs_enumeration(P, K, D, T):
    cfg, lay, lam, X, _ = case(P, K, D, T, seed=P * 1000 + K * 100 + D * 10 + T, trans_ftrs=True)
    L = P * K
    S, M = orc.seg_scores(cfg, lay, lam, X, T)
    bf = orc.brute_force(S, M, T, L, D, K=K)
    rc, ad, al, apt, zx = orc.seg_forward(cfg, S, M, T)
    assert rc == 0 and abs(zx - bf["Zx"]) < 1e-12 * max(1, abs(zx))
    rc, g, xi, zx2 = orc.seg_posteriors(cfg, S, M, T)
    assert rc == 0 and zx2 == zx
    np.testing.assert_allclose(g, bf["gamma"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(xi, bf["xi"], rtol=0, atol=1e-12)
    # the lattice's paths are the allowed labelled segmentations; its best path is the enumeration's
    arcs, ns, fin = orc.seg_lattice_arcs(cfg, S, M, T)
    cnt = np.zeros(ns); cnt[0] = 1
    order = np.argsort(arcs["src"], kind="stable")   # states are numbered in time order: sources before destinations
    for a in arcs[order]:
        cnt[a["dst"]] += cnt[a["src"]]
    assert cnt[fin] == bf["n_paths"]
    ol, cost = orc.best_path(arcs, ns, fin)
    assert abs(-cost - bf["best"][0]) < 1e-4 * max(1.0, abs(bf["best"][0]))
    scores = sorted(p[0] for p in bf["paths"])
    if len(scores) == 1 or scores[-1] - scores[-2] > 1e-4:
        assert [int(x) for x in ol] == [l + L * (d - 1) for (_, d, l) in bf["best"][1]]


@pytest.mark.parametrize("P,K,D,T,tf", [(3, 2, 3, 9, False), (2, 3, 2, 8, True), (3, 3, 4, 12, True)])
def test_gradient_is_derivative_of_loglik(P, K, D, T, tf):
    cfg, lay, lam, X, labs = case(P, K, D, T, seed=7 + T, trans_ftrs=tf)
    rc, grad, numer, zx = orc.seg_build_gradient(cfg, lay, lam, X, labs, T)
    assert rc == 0 and numer <= zx + 1e-9
    rng = np.random.RandomState(2)
    idx = rng.choice(lay.lambda_len, size=min(60, lay.lambda_len), replace=False)
    eps = 1e-6
    for i in idx:
        lp = lam.copy(); lp[i] += eps
        lm = lam.copy(); lm[i] -= eps
        _, _, n1, z1 = orc.seg_build_gradient(cfg, lay, lp, X, labs, T)
        _, _, n0, z0 = orc.seg_build_gradient(cfg, lay, lm, X, labs, T)
        fd = ((n1 - z1) - (n0 - z0)) / (2 * eps)
        assert abs(fd - grad[i]) < 2e-6 * max(1.0, abs(grad[i])), (i, fd, grad[i])


@pytest.mark.parametrize("P,K,D,T,tf,conform", [(3, 2, 3, 10, False, True), (2, 3, 2, 9, True, True), (3, 2, 3, 10, True, False)])
def test_equals_masked_one_state_model(P, K, D, T, tf, conform):
    """The compact n-state model is the dense one with log-0 biases: same partition function, same gradient on the
    weights the topology keeps.  A labelled transition the topology lacks adds nothing to the numerator or the
    gradient (it matches none of the node's transition terms): the dense twin is given the labels without it."""
    cfg, lay, lam, X, labs = case(P, K, D, T, seed=31 + T, trans_ftrs=tf, conform=conform)
    L = P * K
    rc, grad, numer, zx = orc.seg_build_gradient(cfg, lay, lam, X, labs, T)
    assert rc == 0
    dcfg, dlay, dl, s2d = dense_twin(cfg, lay, lam)
    S, M = orc.seg_scores(cfg, lay, lam, X, T)
    dS, dM = orc.seg_scores(dcfg, dlay, dl, X, T)
    rc, _, _, _, dzx = orc.seg_forward(dcfg, dS, dM, T)
    assert rc == 0 and abs(dzx - zx) < 1e-11 * max(1, abs(zx))
    # numerator by hand: state terms + the transitions the topology has
    ends = [t for t in range(T) if labs[t] != 0xffffffff]
    tot = 0.0
    for i, t in enumerate(ends):
        lab = int(labs[t]); al, d = lab % L, lab // L + 1
        tot += S[orc.seg_base(t, D) + d - 1, al]
        if i + 1 < len(ends):
            nl = int(labs[ends[i + 1]]) % L
            if orc.ns_allowed(K, al, nl):
                tot += M[t + 1, al * L + nl]
    assert abs(tot - numer) < 1e-11 * max(1, abs(tot))
    if conform:
        rc, dgrad, dnumer, _ = orc.seg_build_gradient(dcfg, dlay, dl, X, labs, T)
        assert rc == 0 and abs(dnumer - numer) < 1e-11 * max(1, abs(numer))
        np.testing.assert_allclose(dgrad[s2d], grad, rtol=1e-10, atol=1e-12)
    else:
        assert any(not orc.ns_allowed(K, int(labs[a]) % L, int(labs[b]) % L) for a, b in zip(ends, ends[1:]))


def test_lattice_arc_order_and_counts():
    P, K, D, T = 3, 2, 2, 4
    cfg, lay, lam, X, _ = case(P, K, D, T, seed=3)
    L = P * K
    S, M = orc.seg_scores(cfg, lay, lam, X, T)
    arcs, ns, fin = orc.seg_lattice_arcs(cfg, S, M, T)
    assert len(arcs) == (T - 1) * (P * P + 2 * L - P) + orc.num_segs(T, D) * L + L
    assert ns == 1 + L + (T - 1) * 2 * L + 1
    # first boundary node (t = 1): a start state takes the end states ascending and then itself, another state the one
    # before and then itself; sources are node 0's states 1 .. L
    a = arcs[L:]        # node 0 has L arcs (duration 1 from the start state)
    k = 0
    for lab in range(L):
        srcs = ([e for e in range(K - 1, L, K)] + [lab]) if lab % K == 0 else [lab - 1, lab]
        for p in srcs:
            assert a[k]["src"] == 1 + p and a[k]["ilabel"] == 0 and a[k]["dst"] == 1 + L + lab
            assert a[k]["w"] == np.float32(-M[1, p * L + lab])
            k += 1

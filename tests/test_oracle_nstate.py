"""Oracle self-consistency for SURVEY row f3, third step: the n-state frame model (crf_states = K > 1:
nodes/CRF_StdNStateNode.cpp with the sparse transition layout of ftrmaps/CRF_StdFeatureMap.cpp:280-407 and
decoders/CRF_LatticeBuilder.h nStateBuildLattice).  PARITY UNPINNED against the reference binary; cross-checked by
brute-force enumeration of the label sequences the topology allows, central finite differences, and the layout's own
counting identity (numFtrFuncs = nLabs*nsf + (P^2 + 2 nLabs - P)*ntf)."""
import numpy as np
import pytest

import orc


def _case(P, K, T, F, seed=0, scale=0.4, trans_ftrs=True):
    rng = np.random.RandomState(seed)
    X = rng.random_sample((T, F)).astype(np.float32)
    L = P * K
    cfg = orc.config(model_type=orc.STDFRAME, L=L, D=1, F=F, use_trans_ftrs=trans_ftrs, tfs=0, tfe=F - 1, num_states=K)
    lay = orc.Layout(cfg)
    lam = rng.normal(0, scale, lay.lambda_len)
    # a label sequence the topology allows: stay, advance, or (from an end state) jump to a start state
    labs = np.zeros(T, dtype=np.uint32)
    c = int(rng.randint(0, L))
    for t in range(T):
        labs[t] = c
        r = rng.rand()
        if r < 0.4:
            pass
        elif (c + 1) % K == 0:
            c = int(rng.randint(0, P)) * K
        else:
            c = c + 1
    return cfg, lay, lam, X, labs


@pytest.mark.parametrize("P,K", [(2, 2), (3, 3), (4, 2), (2, 5)])
def test_layout_blocks_and_forbidden_transitions(P, K):
    F = 3
    cfg = orc.config(model_type=orc.STDFRAME, L=P * K, D=1, F=F, use_trans_ftrs=True, tfs=0, tfe=F - 1, num_states=K)
    lay = orc.Layout(cfg)
    L = P * K
    nsf, ntf = F + 1, F + 1
    assert lay.lambda_len == L * nsf + (P * P + 2 * L - P) * ntf
    used = np.zeros(lay.lambda_len, dtype=int)
    for c in range(L):
        used[lay.state_idx[c]:lay.state_idx[c] + nsf] += 1
        for p in range(L):
            ti = lay.trans_idx[p * L + c]
            allowed = (p == c) or (c % K == 0 and (p + 1) % K == 0) or (c % K != 0 and p == c - 1)
            assert (ti != 0xffffffff) == allowed, (p, c)
            if allowed:
                used[ti:ti + ntf] += 1
    assert np.all(used == 1)          # every weight belongs to exactly one feature function


@pytest.mark.parametrize("P,K,T", [(2, 2, 4), (3, 2, 5), (2, 3, 6), (3, 3, 4), (2, 2, 1)])
def test_forward_backward_vs_enumeration(P, K, T):
    cfg, lay, lam, X, _ = _case(P, K, T, 3, seed=P * 100 + K * 10 + T)
    S, TD, TO, TE = orc.nstate_scores(cfg, lay, lam, X, T)
    bf = orc.brute_force_nstate(cfg, S, TD, TO, TE, T)
    rc, al, zx = orc.nstate_forward(cfg, S, TD, TO, TE, T)
    assert rc == 0 and abs(zx - bf["Zx"]) < 1e-12 * max(1, abs(zx))
    rc, be = orc.nstate_backward(cfg, S, TD, TO, TE, T)
    assert rc == 0
    np.testing.assert_allclose(np.exp(al + be - zx), bf["gamma"], rtol=0, atol=1e-12)
    # lattice: its paths are the allowed label sequences
    arcs, ns, fin = orc.nstate_lattice_arcs(cfg, S, TD, TO, TE, T)
    cnt = np.zeros(ns); cnt[0] = 1
    for a in arcs:
        cnt[a["dst"]] += cnt[a["src"]]
    assert cnt[fin] == bf["n_paths"]
    ol, cost = orc.best_path(arcs, ns, fin)
    assert abs(-cost - bf["best"][0]) < 1e-4 * max(1.0, abs(bf["best"][0]))
    scores = sorted(p[0] for p in bf["paths"])
    if len(scores) == 1 or scores[-1] - scores[-2] > 1e-4:
        assert list(ol) == list(bf["best"][1])


@pytest.mark.parametrize("P,K,T,tf", [(3, 2, 7, True), (2, 3, 6, False), (4, 3, 9, True)])
def test_gradient_is_derivative_of_loglik(P, K, T, tf):
    cfg, lay, lam, X, labs = _case(P, K, T, 3, seed=5 + T, trans_ftrs=tf)
    rc, grad, numer, zx = orc.nstate_build_gradient(cfg, lay, lam, X, labs, T)
    assert rc == 0 and numer <= zx + 1e-9
    rng = np.random.RandomState(1)
    idx = rng.choice(lay.lambda_len, size=min(60, lay.lambda_len), replace=False)
    eps = 1e-6
    for i in idx:
        lp = lam.copy(); lp[i] += eps
        lm = lam.copy(); lm[i] -= eps
        _, _, n1, z1 = orc.nstate_build_gradient(cfg, lay, lp, X, labs, T)
        _, _, n0, z0 = orc.nstate_build_gradient(cfg, lay, lm, X, labs, T)
        fd = ((n1 - z1) - (n0 - z0)) / (2 * eps)
        assert abs(fd - grad[i]) < 2e-6 * max(1.0, abs(grad[i])), (i, fd, grad[i])
    # the numerator is the score of the labelled sequence
    S, TD, TO, TE = orc.nstate_scores(cfg, lay, lam, X, T)
    tot = sum(S[t, labs[t]] for t in range(T)) + sum(orc.nstate_trans(cfg, TD, TO, TE, t, int(labs[t - 1]), int(labs[t])) for t in range(1, T))
    assert abs(tot - numer) < 1e-11 * max(1, abs(tot))

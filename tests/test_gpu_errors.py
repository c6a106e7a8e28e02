"""-m gpu: error behaviour of the training path through the C ABI.

The reference throws std::runtime_error from computeExpF (bad label, LogMath overflow / log(0),
posterior-mass self-checks, nodes/CRF_StdSegStateNode_WithoutDurLab_WithoutSegTransFtr.cpp:627-631,
:917-947) and the trainer dies with it.  Here scrf_fb_batch itself reports the first failed utterance --
also in the asynchronous form without numer / zx -- and a failed batch leaves the gradient untouched.
The scaled linear-domain recursion of the training path is redone in the log domain when it gives up."""
import numpy as np
import pytest

import orc
import scrf_amd
from cases import Case

pytestmark = pytest.mark.gpu


def bad_labels(c):
    labs = [l.copy() for l in c.labels]
    labs[1][-1] = c.L * c.D + 3           # label >= nActualLabs * labMaxDur
    return labs


@pytest.mark.parametrize("prec", [0, 1])
@pytest.mark.parametrize("want_scalars", [True, False])
def test_bad_label_is_reported_by_fb_batch_and_leaves_the_gradient_alone(prec, want_scalars):
    c = Case(L=5, D=4, in_w=3, Ts=[9, 12, 7], seed=8, precision=prec)
    eng = c.engine()
    good = c.batch(eng)
    eng.zero_grad()
    eng.fb_batch(good, want_scalars=False)
    g0 = eng.get_grad(); s0 = eng.batch_sums()
    bad = eng.batch_from_frames(c.frames, bad_labels(c), c.recipes)
    with pytest.raises(scrf_amd.ScrfError) as ei:
        eng.fb_batch(bad, want_scalars=want_scalars)
    assert ei.value.code == 5 and "utterance 1" in str(ei.value)
    # the failed batch contributed nothing: gradient and sums are those of the good batch
    assert np.array_equal(eng.get_grad(), g0) and np.array_equal(eng.batch_sums(), s0)
    # the engine keeps working, and nothing is sticky
    eng.fb_batch(good, want_scalars=False)
    np.testing.assert_allclose(eng.get_grad(), 2 * g0, rtol=1e-12, atol=1e-12 * np.abs(g0).max())
    eng.synchronize(); eng.sgd_step(0.1)
    assert np.isfinite(eng.get_lambda()).all()
    bad.close(); good.close(); eng.close()


@pytest.mark.parametrize("prec", [0, 1])
def test_the_lowest_failing_utterance_is_the_one_reported(prec):
    """several bad utterances in one batch: the report names the first in batch order every time, as the reference's
    per-utterance loop would have thrown there -- not whichever thread latched first"""
    n = 600
    c = Case(L=5, D=4, in_w=3, Ts=[5 + (u % 4) for u in range(n)], seed=21, precision=prec)
    labs = [l.copy() for l in c.labels]
    for u in (577, 123, 124, 400, 599):
        labs[u][-1] = c.L * c.D + 1
    eng = c.engine()
    bad = eng.batch_from_frames(c.frames, labs, c.recipes)
    for _ in range(5):
        with pytest.raises(scrf_amd.ScrfError) as ei:
            eng.fb_batch(bad, want_scalars=False)
        assert ei.value.code == 5 and "utterance 123" in str(ei.value)
    bad.close(); eng.close()


def wide_spread_case(prec, fused=True):
    """Weights under which the scaled linear-domain recursion must give up where the reference's log-domain
    recursion succeeds: the state bias of label 0 is +1000 (every frame's posterior mass sits on label 0
    to 1000 nats) and every transition OUT of label 0 costs 1500 nats, so the whole transition row of the
    only label that carries mass lies more than 700 nats below the matrix maximum."""
    c = Case(L=4, D=3, in_w=3, Ts=[6, 9, 5], seed=17, precision=prec, lam_scale=0.1)
    lay = c.olay
    nsf = lay.num_state_funcs
    lam = c.lam.copy()
    lam[lay.state_idx[0] + nsf - 1] = 1000.0           # state bias of label 0
    for n in range(c.L):                               # transitions 0 -> n
        lam[lay.trans_idx[0 * c.L + n]] = -1500.0
    c.lam = lam
    return c


@pytest.mark.parametrize("prec", [0, 1])
def test_scores_spanning_more_than_700_nats_fall_back_to_the_log_domain(prec):
    c = wide_spread_case(prec)
    og, on, oz = c.oracle_gradient()          # the oracle (log domain, LogMath) has no trouble
    assert np.isfinite(og).all() and np.isfinite(oz).all()
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    assert eng.train_stats() == 1             # redone once through the log-domain kernels
    g = eng.get_grad()
    assert np.abs(numer - on).max() <= 1e-11 * np.abs(on).max()
    assert np.abs(zx - oz).max() <= 1e-11 * np.abs(oz).max()
    assert np.abs(g - og).max() <= 1e-9 * np.abs(og).max()
    # asynchronous form: same result, counted again
    eng.zero_grad()
    eng.fb_batch(b, want_scalars=False)
    assert eng.train_stats() == 2
    assert np.abs(eng.get_grad() - og).max() <= 1e-9 * np.abs(og).max()
    b.close(); eng.close()


def test_log_domain_overflow_is_an_error_like_the_reference():
    """exp argument above log(DBL_MAX): the reference's expE throws (utils/CRF_LogMath.cpp:211-224); the
    oracle returns its NUMERIC code and so does the engine (after the log-domain retry)."""
    c = Case(L=3, D=2, in_w=2, Ts=[5, 4], seed=3)
    c.lam = c.lam * 1e308                      # scores overflow to +-inf
    rc = orc.seg_build_gradient(c.ocfg, c.olay, c.lam, c.windows(0), c.labels[0], c.Ts[0])[0]
    assert rc != 0
    eng = c.engine(); b = c.batch(eng)
    eng.zero_grad()
    with pytest.raises(scrf_amd.ScrfError) as ei:
        eng.fb_batch(b, want_scalars=False)
    assert ei.value.code == 4
    assert not eng.get_grad().any()
    b.close(); eng.close()


@pytest.mark.parametrize("kw", [dict(L=6, D=4, in_w=3, Ts=[9, 14, 3]),                       # fused, k_post_z
                                dict(L=5, D=3, in_w=3, Ts=[7, 8], trans_ctx=1),              # transition features, k_post_lin
                                dict(L=70, D=3, in_w=4, Ts=[5, 9]),                          # k_dp_lin_mw, two output groups
                                dict(L=6, D=1, in_w=3, Ts=[4, 9], trans_ctx=0, frame_model=True)])
def test_posterior_mass_checks_pass_on_sound_inputs_in_every_recursion(kw, monkeypatch):
    """the reference's self-checks run inside every forward-backward; on sound inputs they must stay
    silent in the linear-domain kernels, the log-domain wavefront kernels and the workgroup kernel"""
    for env in ({}, {"SCRF_LINDP": "0"}):
        for k in ("SCRF_LINDP",):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for prec in (0, 1):
            c = Case(seed=23, precision=prec, **kw)
            eng = c.engine(); b = c.batch(eng)
            eng.fb_batch(b, want_scalars=False)
            assert eng.train_stats() == 0
            og, on, oz = c.oracle_gradient()
            assert np.abs(eng.get_grad() - og).max() <= 1e-9 * np.abs(og).max()
            b.close(); eng.close()


def test_node_value_hook_falls_back_to_the_workgroup_recursion():
    """scrf_forward_backward (getAlpha / getBeta / computeAlphaSum) on the same wide-spread weights: the
    wavefront recursion gives up, the hook answers from the workgroup kernel, equal to the oracle."""
    c = wide_spread_case(0)
    eng = c.engine(); b = c.batch(eng)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        rc, ad, al, apt, zx = orc.seg_forward(c.ocfg, So, Mo, T)
        rc2, be, sd = orc.seg_backward(c.ocfg, So, Mo, T)
        assert rc == 0 and rc2 == 0
        gad, gal, gbe, gzx = eng.forward_backward(b, u, T)
        assert abs(gzx - zx) <= 1e-11 * abs(zx)
        np.testing.assert_allclose(gal, al, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(gbe, be, rtol=1e-11, atol=1e-9)
    b.close(); eng.close()

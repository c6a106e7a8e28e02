"""-m gpu: every BASELINE.json config at its stated model shape, through the C ABI, against the oracle.

  config 2: tests/test_gpu_parity.py::test_full_size_config2_utterances
  config 3: TIMIT demo shape -- 48 labels, max duration 10, 144-dim posterior-like stream (segment
            features, 1162 per window) + the same frames with +-6 context (1872 transition features),
            `stdtrans` map, lambda_len 4,371,216 (demo/segmental-timit-demo.cfg.in)
  config 4: lattice decode of a TRAINED config-3 model (weights after SGD steps on the device):
            arcs byte for byte, best path labels and float cost
  config 5: stress shape -- 200 labels, max duration 40, 123-dim frames; T = 300 against the oracle,
            T = 2000 through size-independent properties (mixed forward-backward + Viterbi)
Bars: Zx / numerator / gradient vs oracle 1e-10 (EXACT) and 1e-9 (FAST) relative -- the contract is
1e-4 (BASELINE.json); arcs, Viterbi labels and costs bit-identical."""
import numpy as np
import pytest

import orc
import scrf_amd
from cases import Case

pytestmark = pytest.mark.gpu

CFG3 = dict(L=48, D=10, in_w=144, trans_ctx=6, lam_scale=0.3, l1_norm=True)
CFG5 = dict(L=200, D=40, in_w=123, lam_scale=0.01)


def n_arcs_seg(T, L, D):
    return (T - 1) * L * L + orc.num_segs(T, D) * L + L


@pytest.mark.parametrize("prec,tol", [(0, 1e-10), (1, 1e-9)], ids=["exact", "fast"])
def test_config3_timit_demo_shape_forward_backward(prec, tol):
    c = Case(Ts=[120, 304, 200], seed=3, precision=prec, **CFG3)
    assert c.F == 8 * 144 + 10 + 13 * 144 and c.olay.lambda_len == 4371216
    eng = c.engine(); b = c.batch(eng)
    assert eng.lambda_len == 4371216 and eng.num_state_funcs() == 1163 and eng.num_trans_funcs() == 1873
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, on, oz = c.oracle_gradient()
    assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - oz).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() <= tol * np.abs(og).max()
    s = eng.batch_sums()
    assert s[2] == 3 and abs(s[1] - oz.sum()) <= tol * abs(oz.sum())
    b.close(); eng.close()


def test_config3_lattice_arcs_and_viterbi():
    c = Case(Ts=[304, 77], seed=13, **CFG3)
    eng = c.engine(); b = c.batch(eng, with_labels=False)
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        assert len(oa) == n_arcs_seg(T, c.L, c.D)          # 841,920 arcs at T = 304
        ga, gns, gfin = eng.lattice_arcs(b, u)
        assert (gns, gfin) == (ons, ofin) and gns == 2 * c.L * T - c.L + 2
        assert ga.tobytes() == oa.tobytes()
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol)
        assert np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()
    b.close(); eng.close()


def test_config4_lattice_decode_of_a_trained_model():
    """config 4: the config-3 model after SGD steps on the device (the weights CRFFstDecode would read),
    batched best paths + arc lists against the oracle's lattice and shortest path under the SAME weights."""
    c = Case(Ts=[150, 304, 90, 211, 64, 180], seed=4, precision=1, **CFG3)
    eng = c.engine(); b = c.batch(eng)
    for _ in range(3):                       # three minibatch steps: lambda moves away from its random start
        eng.zero_grad()
        eng.fb_batch(b, want_scalars=False)
        eng.sgd_step(0.1 / len(c.Ts), False)
    lam = eng.get_lambda()
    assert np.isfinite(lam).all() and np.abs(lam - c.lam).max() > 1e-4
    labs, cost = eng.viterbi_batch(b)
    for u in (1, 4):
        T = c.Ts[u]
        So, Mo = orc.seg_scores(c.ocfg, c.olay, lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        ga, gns, gfin = eng.lattice_arcs(b, u)
        assert (gns, gfin) == (ons, ofin) and ga.tobytes() == oa.tobytes()
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol) and np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()
    for u, T in enumerate(c.Ts):             # every path covers its utterance with durations <= D
        durs = [int(x) // c.L + 1 for x in labs[u]]
        assert sum(durs) == T and max(durs) <= c.D
    b.close(); eng.close()


@pytest.mark.parametrize("prec,tol", [(0, 1e-10), (1, 1e-9)], ids=["exact", "fast"])
def test_config5_stress_shape_against_the_oracle(prec, tol):
    c = Case(Ts=[300, 97], seed=5, precision=prec, **CFG5)
    assert c.F == 8 * 123 + 40 and c.olay.lambda_len == 245000
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, on, oz = c.oracle_gradient()
    assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - oz).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() <= tol * np.abs(og).max()
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol) and np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()
    b.close(); eng.close()


def test_config5_full_length_utterances_properties():
    """config 5 at T = 2000 (79,220 windows x 200 labels per utterance): half of the batch through
    forward-backward, half through Viterbi, as the config says; checked through properties that do not
    need the oracle at this size, plus the oracle's Zx on the shortest utterance."""
    c = Case(Ts=[2000, 2000, 2000, 400], seed=55, precision=1, **CFG5)
    eng = c.engine()
    fb = eng.batch_from_frames(c.frames[:2] + c.frames[3:], c.labels[:2] + c.labels[3:], c.recipes)
    vb = eng.batch_from_frames(c.frames[2:], None, c.recipes)
    numer, zx = eng.fb_batch(fb)
    g = eng.get_grad()
    assert np.isfinite(g).all() and np.isfinite(zx).all() and (numer < zx).all()
    L, F = c.L, c.F
    nsf = F + 1
    stride = nsf + L
    # sum of the state-bias gradients = #true segments - E[#segments]; of the transition-bias gradients
    # = the same minus one per utterance on both sides -> the two sums agree
    sb = sum(g[l * stride + F] for l in range(L))
    tb = sum(g[l * stride + nsf:(l + 1) * stride].sum() for l in range(L))
    assert abs(sb - tb) < 1e-6 * sum(c.Ts)
    # one-hot duration features: column 8W + d - 1 fires once per window of length d, so the gradient summed
    # over labels and durations equals the state-bias sum (every window has exactly one duration)
    W = c.in_w
    dsum = sum(g[l * stride + 8 * W + d] for l in range(L) for d in range(c.D))
    assert abs(dsum - sb) < 1e-6 * sum(c.Ts)
    # Zx of the short utterance against the oracle's forward pass
    T = c.Ts[3]
    So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(3), T)
    rc, ad, al, apt, ozx = orc.seg_forward(c.ocfg, So, Mo, T)
    assert rc == 0 and abs(zx[2] - ozx) <= 1e-11 * abs(ozx)
    labs, cost = eng.viterbi_batch(vb)
    for u, T in enumerate(c.Ts[2:]):
        durs = [int(x) // L + 1 for x in labs[u]]
        assert sum(durs) == T and max(durs) <= c.D and np.isfinite(cost[u])
    # the best path's cost can not exceed -log of any other path's score: compare with the labelled path
    oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
    ol, oc = orc.best_path(oa, ons, ofin)
    assert list(labs[1]) == list(ol) and np.float32(cost[1]).tobytes() == np.float32(oc).tobytes()
    fb.close(); vb.close(); eng.close()


def test_model_type_stdseg_no_dur_no_transftr():
    """`stdseg_no_dur_no_transftr` -- the segmental node without transition features at all.  Served by the
    same engine as the TIMIT-demo model with bias-only transitions: the gradient, the scores and the best path are
    the oracle's and bit-identical to model type `..._no_segtransftr` under a `stdstate` map; a `stdtrans` map is
    refused, as the reference's main does (CRFTrain/src/Main.cpp:465-468)."""
    import orc
    import scrf_amd
    from cases import Case
    res = {}
    for mt in (orc.STDSEG_NO_DUR_NO_TRANSFTR, orc.STDSEG_NO_DUR_NO_SEGTRANSFTR):
        c = Case(L=7, D=5, in_w=4, Ts=[11, 4, 17], seed=33, model_type=mt, precision=0)
        eng = c.engine(); b = c.batch(eng)
        numer, zx = eng.fb_batch(b)
        g = eng.get_grad()
        labs, cost = eng.viterbi_batch(b)
        og, on, oz = c.oracle_gradient()
        assert np.abs(g - og).max() <= 1e-9 * np.abs(og).max()
        assert np.abs(zx - oz).max() <= 1e-11 * np.abs(oz).max() and np.abs(numer - on).max() <= 1e-11 * max(1.0, np.abs(on).max())
        res[mt] = (g.tobytes(), numer.tobytes(), zx.tobytes(), [list(l) for l in labs], cost.tobytes())
        b.close(); eng.close()
    assert res[orc.STDSEG_NO_DUR_NO_TRANSFTR] == res[orc.STDSEG_NO_DUR_NO_SEGTRANSFTR]
    # stdtrans with this model type: refused at creation with the reference's message
    c = Case(L=7, D=5, in_w=4, Ts=[6], seed=1, trans_ctx=1, model_type=orc.STDSEG_NO_DUR_NO_TRANSFTR)
    with pytest.raises(scrf_amd.ScrfError) as ei:
        scrf_amd.Engine(c.gcfg)
    assert ei.value.code == 1 and 'must be "stdstate"' in str(ei.value)

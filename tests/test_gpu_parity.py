"""-m gpu: parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bars: bit-exact for window vectors, scores (EXACT mode), lattice arcs and Viterbi label
sequences; forward log-likelihood and gradients within 1e-4 relative (BASELINE.json) -- the
tests assert much tighter bounds than the contract wherever the fp64 path allows it."""
import os

import numpy as np
import pytest

import orc
import scrf_amd
from cases import Case

pytestmark = pytest.mark.gpu

REL_CONTRACT = 1e-4  # BASELINE.json north_star: log-likelihood and gradients within 1e-4 relative


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64 if a.dtype == np.float64 else np.uint32)


CASES = [
    dict(L=3, D=3, in_w=2, Ts=[1, 2, 3, 4, 7]),                       # T around D, incl. T=1
    dict(L=2, D=4, in_w=3, Ts=[3, 4, 5, 12], trans_ctx=1),            # transition features
    dict(L=5, D=1, in_w=4, Ts=[1, 6, 9], trans_ctx=2),                # D=1 segmental
    dict(L=7, D=10, in_w=5, Ts=[9, 10, 11, 30]),                      # T = D-1, D, D+1, 3D
    dict(L=48, D=25, in_w=39, Ts=[60, 33]),                           # config-2 shape, short
    dict(L=48, D=10, in_w=8, Ts=[40, 25], trans_ctx=1, lam_scale=0.05),  # TIMIT-like: stdtrans
]


@pytest.fixture(scope="module", params=range(len(CASES)), ids=lambda i: "case%d" % i)
def case(request):
    c = Case(seed=100 + request.param, **CASES[request.param])
    eng = c.engine()
    b = c.batch(eng)
    yield c, eng, b
    b.close(); eng.close()


def test_layout_hooks(case):
    c, eng, _ = case
    assert eng.lambda_len == c.olay.lambda_len
    assert eng.num_state_funcs() == c.olay.num_state_funcs and eng.num_trans_funcs() == c.olay.num_trans_funcs
    for l in range(c.L):
        assert eng.state_idx(l) == c.olay.state_idx[l]
        for p in range(c.L):
            assert eng.trans_idx(p, l) == c.olay.trans_idx[p * c.L + l]


def test_windows_bit_exact(case):
    c, eng, b = case
    for u, T in enumerate(c.Ts):
        assert np.array_equal(bits(eng.windows(b, u, T)), bits(c.windows(u)))


def test_scores_bit_exact(case):
    c, eng, b = case
    for u, T in enumerate(c.Ts):
        S, M = eng.scores(b, u, T)
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        assert np.array_equal(bits(S), bits(So))
        assert np.array_equal(bits(M), bits(Mo))


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


def test_fb_batch_gradient(case):
    c, eng, b = case
    eng.zero_grad()
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, onumer, ozx = c.oracle_gradient()
    assert np.abs(numer - onumer).max() <= 1e-12 * max(1, np.abs(onumer).max())
    assert np.abs(zx - ozx).max() <= 1e-11 * np.abs(ozx).max()
    scale = np.abs(og).max()
    err = np.abs(g - og).max() / scale
    assert err <= REL_CONTRACT
    assert err <= 1e-9, err  # what the fp64 path actually delivers
    s = eng.batch_sums()
    assert abs(s[0] - onumer.sum()) < 1e-9 * max(1, abs(onumer.sum())) and s[2] == len(c.Ts)
    # gradient accumulates (+=) like the reference's per-thread buffer
    eng.fb_batch(b)
    np.testing.assert_allclose(eng.get_grad(), 2 * g, rtol=1e-12, atol=1e-12 * scale)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_fb_batch_gradient_fast_precision(ci):
    """FAST training precision: scores and expected counts on fp64 MFMA (sums reordered)."""
    c = Case(seed=100 + ci, precision=1, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, onumer, ozx = c.oracle_gradient()
    assert np.abs(numer - onumer).max() <= 1e-11 * max(1, np.abs(onumer).max())
    assert np.abs(zx - ozx).max() <= 1e-11 * np.abs(ozx).max()
    err = np.abs(g - og).max() / np.abs(og).max()
    assert err <= REL_CONTRACT and err <= 1e-9, err
    # decode entry points stay EXACT whatever the training precision
    So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(0), c.Ts[0])
    S, M = eng.scores(b, 0, c.Ts[0])
    assert np.array_equal(bits(S), bits(So)) and np.array_equal(bits(M), bits(Mo))
    b.close(); eng.close()


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_fb_batch_gradient_fast32_precision(ci):
    """FAST32 (opt-in): both dense contractions on the f32 MFMA, f64 everywhere else.  The bound
    asserted here (1e-5) is what demonstrates the 1e-4 contract for this mode."""
    c = Case(seed=100 + ci, precision=2, **CASES[ci])
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, onumer, ozx = c.oracle_gradient()
    assert np.abs(numer - onumer).max() <= 1e-5 * max(1, np.abs(onumer).max())
    assert np.abs(zx - ozx).max() <= 1e-6 * np.abs(ozx).max()
    err = np.abs(g - og).max() / np.abs(og).max()
    assert err <= REL_CONTRACT and err <= 1e-5, err
    b.close(); eng.close()


FASTLIN_SHAPES = [
    dict(L=48, D=25, in_w=39, Ts=[300, 57, 24, 1]),     # config-2 shape, utterances shorter than D
    dict(L=3, D=3, in_w=2, Ts=[1, 2, 3, 4, 7]),
    dict(L=7, D=10, in_w=5, Ts=[9, 10, 11, 30]),
    dict(L=64, D=12, in_w=20, Ts=[40, 5]),              # the widest label set the tier takes
    dict(L=20, D=40, in_w=9, Ts=[100, 41, 39]),         # the longest duration
    dict(L=33, D=7, in_w=45, Ts=[50]),                  # two column chunks per dense group, odd label count
]


@pytest.mark.parametrize("si", range(len(FASTLIN_SHAPES)))
def test_fb_batch_gradient_fastlin_precision(si):
    """FASTLIN (what bench.py runs): FAST with the window average taken as the exact mean -- linear in the frames, so it
    leaves both dense contractions (prefix sums of a per-frame projection; a sixth per-frame sum of R).  The feature
    values differ from the reference's float arithmetic by its rounding, so this is a tolerance of its own: 1e-6
    relative on the gradient against the oracle (the reference's floats), contract 1e-4."""
    c = Case(seed=700 + si, precision=scrf_amd.PREC_FASTLIN, **FASTLIN_SHAPES[si])
    eng = c.engine(); b = c.batch(eng)
    assert eng.batch_fused_mode(b) == 2
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, onumer, ozx = c.oracle_gradient()
    assert np.abs(numer - onumer).max() <= 1e-6 * max(1, np.abs(onumer).max())
    assert np.abs(zx - ozx).max() <= 1e-8 * np.abs(ozx).max()
    err = np.abs(g - og).max() / np.abs(og).max()
    assert err <= REL_CONTRACT and err <= 1e-6, err
    # a second call accumulates the same gradient again (slabs, rings and prefix sums are rebuilt per call)
    eng.fb_batch(b)
    assert np.abs(eng.get_grad() - 2 * g).max() <= 1e-12 * np.abs(g).max()
    # decode entry points stay EXACT whatever the training precision
    So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(0), c.Ts[0])
    S, M = eng.scores(b, 0, c.Ts[0])
    assert np.array_equal(bits(S), bits(So)) and np.array_equal(bits(M), bits(Mo))
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol)
    b.close(); eng.close()


@pytest.mark.parametrize("shape", [dict(L=5, D=4, in_w=3, Ts=[1, 2, 3, 4, 5, 9, 30]), dict(L=48, D=10, in_w=39, Ts=[120, 9, 10, 11]),
                                   dict(L=13, D=8, in_w=45, Ts=[70, 3])])
def test_fastlin_is_fast_when_the_float_average_is_exact(shape):
    """Frame values on a grid where every float running sum and its quotient by the window length are exact: the
    reference's float average equals the exact mean, and FASTLIN has to agree with the oracle as tightly as FAST does
    (1e-9) -- which pins the prefix-sum gather of the score kernel and the ring sums of k_post_z, edges included."""
    c = Case(seed=810, precision=scrf_amd.PREC_FASTLIN, exact_avg=True, **shape)
    eng = c.engine(); b = c.batch(eng)
    assert eng.batch_fused_mode(b) == 2
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, onumer, ozx = c.oracle_gradient()
    assert np.abs(numer - onumer).max() <= 1e-11 * max(1, np.abs(onumer).max())
    assert np.abs(zx - ozx).max() <= 1e-11 * np.abs(ozx).max()
    assert np.abs(g - og).max() / np.abs(og).max() <= 1e-9
    b.close(); eng.close()


def test_fastlin_falls_back_to_fast_where_its_kernels_do_not_apply():
    """L > 64: the batch runs the FAST kernels under FASTLIN (same contract)."""
    c = Case(seed=820, precision=scrf_amd.PREC_FASTLIN, L=70, D=3, in_w=4, Ts=[5, 9, 14])
    eng = c.engine(); b = c.batch(eng)
    assert eng.batch_fused_mode(b) == 1
    numer, zx = eng.fb_batch(b)
    og, on, oz = c.oracle_gradient()
    assert np.abs(zx - oz).max() <= 1e-11 * np.abs(oz).max()
    assert np.abs(eng.get_grad() - og).max() / np.abs(og).max() <= 1e-9
    b.close(); eng.close()


MIXED_SHAPES = [
    dict(L=6, D=4, in_w=5, Ts=[9, 14, 3, 1], trans_ctx=1),
    dict(L=48, D=10, in_w=8, Ts=[40, 25], trans_ctx=1, lam_scale=0.05),          # TIMIT-like label space
    dict(L=5, D=3, in_w=90, Ts=[12, 7], trans_ctx=1, lam_scale=0.05),            # wide stream: no k_pframe / k_ztf (W > 80), z-blocked count kernel under FAST
    dict(L=48, D=10, in_w=144, Ts=[30, 11], trans_ctx=2, lam_scale=0.02),        # config 3's widths: z-blocked under FASTLIN too
]


@pytest.mark.parametrize("si", range(len(MIXED_SHAPES)))
def test_fused_state_part_with_materialised_transition_streams(si, monkeypatch):
    """BASELINE config 3's structure (round 4): the state features are stream 0's segment-recipe window, the transition
    features a second (context) stream.  FAST / FASTLIN fuse the window synthesis of the STATE part into its two
    contractions and keep the first-row windows + dense contractions for the transition part; SCRF_FUSE_MIXED=0 keeps the
    general path for everything.  Both against the oracle; FAST also against each other (summation order only)."""
    res = {}
    for prec, tol in ((scrf_amd.PREC_FAST, 1e-9), (scrf_amd.PREC_FASTLIN, 1e-6)):
        for mixed in ("1", "0"):
            monkeypatch.setenv("SCRF_FUSE_MIXED", mixed)
            c = Case(seed=830 + si, precision=prec, **MIXED_SHAPES[si])
            eng = c.engine(); b = c.batch(eng)
            assert eng.batch_fused_mode(b) == (0 if mixed == "0" else (2 if prec == scrf_amd.PREC_FASTLIN else 1))
            numer, zx = eng.fb_batch(b)
            g = eng.get_grad()
            og, on, oz = c.oracle_gradient()
            assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
            assert np.abs(zx - oz).max() <= max(1e-11, tol * 1e-2) * np.abs(oz).max()
            assert np.abs(g - og).max() / np.abs(og).max() <= tol
            res[(prec, mixed)] = g.copy()
            # decode of such a batch keeps the EXACT path
            labs, cost = eng.viterbi_batch(b)
            So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(0), c.Ts[0])
            oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, c.Ts[0])
            ol, _ = orc.best_path(oa, ons, ofin)
            assert list(labs[0]) == list(ol)
            b.close(); eng.close()
    a, bb = res[(scrf_amd.PREC_FAST, "1")], res[(scrf_amd.PREC_FAST, "0")]
    assert np.abs(a - bb).max() <= 1e-10 * np.abs(bb).max()


HYBRID_SHAPES = [   # (streams too wide for the fused kernels at these label counts: 3 W + D + 1 > 208 columns)
    dict(L=65, D=5, in_w=70, Ts=[1, 2, 9, 30], lam_scale=0.05),      # two 64-output groups, utterances shorter than 2 D
    dict(L=130, D=12, in_w=66, Ts=[27, 1, 260], lam_scale=0.05),     # a long utterance in a small launch: frame segments in k_lin_z5
    dict(L=200, D=40, in_w=60, Ts=[45, 130], lam_scale=0.05),        # config 5's label space and maximum duration
    dict(L=96, D=25, in_w=62, Ts=[60, 24], lam_scale=0.05),
    dict(L=70, D=6, in_w=90, Ts=[20, 7], lam_scale=0.05),            # sampled blocks through the generic contractions (W > 80)
    dict(L=66, D=4, in_w=69, Ts=[9, 14, 3, 1, 11, 8], scratch_bytes=1 << 18, lam_scale=0.05),   # several chunks
]


@pytest.mark.parametrize("si", range(len(HYBRID_SHAPES)))
def test_hybrid_path_sampled_blocks_leave_the_dense_contractions(si, monkeypatch):
    """BASELINE config 5's structure (round 4): one segment-recipe stream with more labels than the fused kernels' LDS
    images take.  The window vectors stay materialised, but the five sampled blocks go through per-frame projections
    (scores: k_add_p) and per-frame sums (counts: k_lin_z5 + Z^T F), the one-hot duration and bias counts are sums of R, and
    the dense contractions keep [avg | max | min] only.  Against the oracle and against the general path (SCRF_HYBRID=0):
    exact re-associations, FAST bounds."""
    res = {}
    for hy in ("1", "0"):
        monkeypatch.setenv("SCRF_HYBRID", hy)
        c = Case(seed=870 + si, precision=scrf_amd.PREC_FAST, **HYBRID_SHAPES[si])
        eng = c.engine(); b = c.batch(eng)
        assert eng.batch_fused_mode(b) == (3 if hy == "1" else 0)
        numer, zx = eng.fb_batch(b)
        g = eng.get_grad()
        og, on, oz = c.oracle_gradient()
        assert np.abs(numer - on).max() <= 1e-9 * max(1, np.abs(on).max())
        assert np.abs(zx - oz).max() <= 1e-11 * np.abs(oz).max()
        assert np.abs(g - og).max() / np.abs(og).max() <= 1e-9
        res[hy] = g.copy()
        b.close(); eng.close()
    assert np.abs(res["1"] - res["0"]).max() <= 1e-10 * np.abs(res["0"]).max()


FUSED_SHAPES = [
    dict(L=3, D=3, in_w=2, Ts=[1, 2, 3, 4, 7]),
    dict(L=50, D=5, in_w=45, Ts=[1, 13, 40, 77]),       # two output groups, two column chunks per group
    dict(L=7, D=32, in_w=5, Ts=[31, 32, 33, 100]),      # D = rows of an expected-count tile
    dict(L=48, D=25, in_w=39, Ts=[300, 57]),            # config-2 shape
]


@pytest.mark.parametrize("si", range(len(FUSED_SHAPES)))
def test_fused_window_synthesis_equals_materialised_windows(si, monkeypatch):
    """FAST / FAST32 rebuild the window vectors inside both contractions (no X in HBM) when the
    input is one segment-recipe stream; SCRF_FUSE=0 forces the materialised-X kernels.  Window
    values are the same floats either way, so the two differ by summation order only."""
    for prec, tol in ((1, 1e-11), (2, 1e-5)):
        res = []
        for fuse in ("1", "0"):
            monkeypatch.setenv("SCRF_FUSE", fuse)
            c = Case(seed=300 + si, precision=prec, **FUSED_SHAPES[si])
            eng = c.engine(); b = c.batch(eng)
            assert eng.batch_is_fused(b) == (fuse == "1")
            numer, zx = eng.fb_batch(b)
            res.append((numer, zx, eng.get_grad()))
            b.close(); eng.close()
        (n1, z1, g1), (n2, z2, g2) = res
        scale = np.abs(g2).max()
        np.testing.assert_allclose(n1, n2, rtol=tol, atol=tol * max(1, np.abs(n2).max()))
        np.testing.assert_allclose(z1, z2, rtol=tol)
        assert np.abs(g1 - g2).max() <= 10 * tol * scale
        og, on, oz = c.oracle_gradient()
        assert np.abs(g1 - og).max() / np.abs(og).max() <= max(1e-9, tol)


@pytest.mark.parametrize("prec", [0, 1, 2])
def test_more_than_64_labels(prec):
    """64 < L <= 256: the multi-wavefront linear-domain recursion (k_dp_lin_mw: one workgroup per
    utterance and direction, the transition operand exchanged through LDS), the 64 x 64-blocked
    transition-count contraction, and the fused contractions with several 48-output groups per row."""
    c = Case(L=70, D=3, in_w=4, Ts=[1, 2, 5, 9, 14], seed=61, precision=prec)
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    og, on, oz = c.oracle_gradient()
    tol = 1e-9 if prec < 2 else 1e-5
    assert np.abs(numer - on).max() <= tol * max(1, np.abs(on).max())
    assert np.abs(zx - oz).max() <= tol * np.abs(oz).max()
    assert np.abs(g - og).max() / np.abs(og).max() <= tol
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol)
    b.close(); eng.close()


def test_more_than_64_labels_log_domain_fallback(monkeypatch):
    """SCRF_LINDP=0 sends L > 64 through the workgroup-per-utterance log-domain recursion (k_fb)."""
    monkeypatch.setenv("SCRF_LINDP", "0")
    c = Case(L=70, D=3, in_w=4, Ts=[1, 2, 5, 9, 14], seed=61, precision=1)
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    og, on, oz = c.oracle_gradient()
    assert np.abs(zx - oz).max() <= 1e-11 * np.abs(oz).max()
    assert np.abs(eng.get_grad() - og).max() / np.abs(og).max() <= 1e-9
    b.close(); eng.close()


def test_lattice_arcs_bit_exact(case):
    c, eng, b = case
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        ga, gns, gfin = eng.lattice_arcs(b, u)
        assert (gns, gfin) == (ons, ofin)
        assert ga.tobytes() == oa.tobytes()


def test_viterbi_matches_shortest_path_on_reference_lattice(case):
    c, eng, b = case
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol)
        assert np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()


@pytest.mark.parametrize("si", range(len(FUSED_SHAPES)))
def test_fast_decode_is_bit_identical_to_exact_decode(si, monkeypatch):
    """scrf_viterbi_batch on raw frames takes float arc weights from the fused fp64-MFMA score kernel
    and recomputes in reference order every weight whose float rounding the error bound cannot
    guarantee; labels and float path costs must equal the EXACT path's (SCRF_FAST_DECODE=0) bit for
    bit -- also when the screen is widened so that many entries take the fix-up kernel
    (SCRF_DECODE_BOUND_SCALE=30), when the list overflows and the chunk falls back to the EXACT
    path (1e7), and when the batch is cut into several chunks."""
    kw = dict(seed=500 + si, lam_scale=0.3, **FUSED_SHAPES[si])
    res = {}
    for tag, env, scratch in [("exact", {"SCRF_FAST_DECODE": "0"}, 0), ("fast", {}, 0), ("fix", {"SCRF_DECODE_BOUND_SCALE": "30"}, 0),
                              ("overflow", {"SCRF_DECODE_BOUND_SCALE": "1e7"}, 0), ("chunks", {"SCRF_DECODE_BOUND_SCALE": "30"}, 1 << 16)]:
        for k in ("SCRF_FAST_DECODE", "SCRF_DECODE_BOUND_SCALE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = Case(scratch_bytes=scratch, **kw)
        eng = c.engine(); b = c.batch(eng, with_labels=False)
        assert eng.batch_is_fused(b)
        labs, cost = eng.viterbi_batch(b)
        res[tag] = (labs, cost, eng.decode_stats())
        b.close(); eng.close()
    el, ec, est = res["exact"]
    assert est == (0, 0)
    n_entries = sum(orc.num_segs(T, c.D) for T in c.Ts) * c.L
    for tag in ("fast", "fix", "overflow", "chunks"):
        gl, gc, st = res[tag]
        assert all(list(a) == list(b_) for a, b_ in zip(gl, el)), tag
        assert gc.tobytes() == ec.tobytes(), tag
    assert res["fast"][2][1] == 0 and res["fast"][2][0] <= max(8, n_entries // 1000)
    assert res["fix"][2][1] == 0 and res["fix"][2][0] >= res["fast"][2][0]
    if n_entries > 8192:
        assert res["fix"][2][0] > max(res["fast"][2][0], n_entries // 100000)   # the widened screen really feeds the fix-up kernel
    if n_entries > 8192:
        assert res["overflow"][2][1] >= 1
    # and the oracle's shortest path on the oracle's lattice
    u = len(c.Ts) - 1; T = c.Ts[u]
    So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
    oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, So, Mo, T)
    ol, oc = orc.best_path(oa, ons, ofin)
    assert list(res["fast"][0][u]) == list(ol) and np.float32(res["fast"][1][u]).tobytes() == np.float32(oc).tobytes()


def test_viterbi_tie_rule_on_zero_weights():
    c = Case(L=3, D=2, in_w=2, Ts=[4, 5], seed=1)
    c.lam[:] = 0.0
    eng = c.engine(); b = c.batch(eng)
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        S = np.zeros((orc.num_segs(T, c.D), c.L)); M = np.zeros((T, c.L * c.L))
        oa, ons, ofin = orc.seg_lattice_arcs(c.ocfg, S, M, T)
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol) and cost[u] == oc
    b.close(); eng.close()


def test_frame_model_config1_shape():
    """frame-level CRF (a15, a17): gradient, lattice arcs and best path."""
    c = Case(L=6, D=1, in_w=3, Ts=[4, 3, 4, 9], trans_ctx=0, seed=5, frame_model=True)
    eng = c.engine(); b = c.batch(eng)
    numer, zx = eng.fb_batch(b)
    og, on, oz = c.oracle_gradient()
    assert np.abs(numer - on).max() < 1e-12 * max(1, np.abs(on).max()) and np.abs(zx - oz).max() < 1e-11 * np.abs(oz).max()
    assert np.abs(eng.get_grad() - og).max() / np.abs(og).max() < 1e-9
    labs, cost = eng.viterbi_batch(b)
    for u, T in enumerate(c.Ts):
        So, Mo = orc.seg_scores(c.ocfg, c.olay, c.lam, c.windows(u), T)
        oa, ons, ofin = orc.frame_lattice_arcs(c.ocfg, So, Mo, T)
        ga, gns, gfin = eng.lattice_arcs(b, u)
        assert (gns, gfin) == (ons, ofin) and ga.tobytes() == oa.tobytes()
        ol, oc = orc.best_path(oa, ons, ofin)
        assert list(labs[u]) == list(ol) and np.float32(cost[u]).tobytes() == np.float32(oc).tobytes()
    b.close(); eng.close()


def test_materialised_windows_input_equals_frame_input():
    c = Case(L=4, D=3, in_w=3, Ts=[5, 8], trans_ctx=1, seed=9)
    eng = c.engine()
    b1 = c.batch(eng)
    b2 = eng.batch_from_windows([c.windows(u) for u in range(2)], c.Ts, c.labels)
    eng.zero_grad(); n1, z1 = eng.fb_batch(b1); g1 = eng.get_grad()
    eng.zero_grad(); n2, z2 = eng.fb_batch(b2); g2 = eng.get_grad()
    assert np.array_equal(n1, n2) and np.array_equal(z1, z2) and np.array_equal(g1, g2)
    b1.close(); b2.close(); eng.close()


def test_chunking_is_invisible():
    """a tiny scratch budget forces one chunk per utterance; results must not change."""
    kw = dict(L=5, D=4, in_w=3, Ts=[6, 9, 4, 12, 7], trans_ctx=1, seed=21)
    c1 = Case(**kw); c2 = Case(scratch_bytes=1 << 16, **kw)
    e1, e2 = c1.engine(), c2.engine()
    b1, b2 = c1.batch(e1), c2.batch(e2)
    n1, z1 = e1.fb_batch(b1); n2, z2 = e2.fb_batch(b2)
    assert np.array_equal(n1, n2) and np.array_equal(z1, z2)
    np.testing.assert_allclose(e1.get_grad(), e2.get_grad(), rtol=1e-12, atol=1e-13)
    l1, c1c = e1.viterbi_batch(b1); l2, c2c = e2.viterbi_batch(b2)
    assert all(list(a) == list(bb) for a, bb in zip(l1, l2)) and np.array_equal(c1c, c2c)
    for x in (b1, b2): x.close()
    e1.close(); e2.close()


def test_fused_path_chunking_and_log_domain_fallback(monkeypatch):
    """the fused FAST path must not depend on how the batch is cut into chunks (chunk-relative tile,
    row and frame offsets), and the scaled linear-domain recursion must agree with the log-domain
    kernels (SCRF_LINDP=0) to rounding."""
    kw = dict(L=6, D=5, in_w=4, Ts=[1, 2, 4, 5, 6, 9, 17, 30, 3, 12], seed=41, precision=1)
    c1 = Case(**kw); c2 = Case(scratch_bytes=1 << 15, **kw)
    e1, e2 = c1.engine(), c2.engine()
    b1, b2 = c1.batch(e1), c2.batch(e2)
    n1, z1 = e1.fb_batch(b1); n2, z2 = e2.fb_batch(b2)
    np.testing.assert_allclose(n1, n2, rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(z1, z2, rtol=1e-13)
    g1, g2 = e1.get_grad(), e2.get_grad()
    np.testing.assert_allclose(g1, g2, rtol=1e-11, atol=1e-12 * np.abs(g1).max())
    og, on, oz = c1.oracle_gradient()
    assert np.abs(g1 - og).max() / np.abs(og).max() <= 1e-9
    assert np.abs(n1 - on).max() <= 1e-11 * max(1, np.abs(on).max()) and np.abs(z1 - oz).max() <= 1e-11 * np.abs(oz).max()
    for x in (b1, b2): x.close()
    e1.close(); e2.close()
    monkeypatch.setenv("SCRF_LINDP", "0")
    c3 = Case(**kw); e3 = c3.engine(); b3 = c3.batch(e3)
    n3, z3 = e3.fb_batch(b3); g3 = e3.get_grad()
    np.testing.assert_allclose(z1, z3, rtol=1e-12)
    np.testing.assert_allclose(n1, n3, rtol=1e-12, atol=1e-12)
    assert np.abs(g1 - g3).max() <= 1e-10 * np.abs(g1).max()
    b3.close(); e3.close()


def test_two_lane_pipeline_equals_single_lane(monkeypatch):
    """batches of >= 64 utterances are cut into >= 4 chunks that alternate between two HIP streams
    (DP of one chunk overlaps the contractions of the other); the result must not depend on it."""
    kw = dict(L=6, D=4, in_w=3, Ts=[5 + (7 * i) % 11 for i in range(80)], seed=77)
    res = []
    for lanes in ("1", "2"):
        monkeypatch.setenv("SCRF_LANES", lanes)
        for prec in (0, 1):
            c = Case(precision=prec, **kw)
            eng = c.engine(); b = c.batch(eng)
            numer, zx = eng.fb_batch(b)
            res.append((numer, zx, eng.get_grad(), eng.batch_sums()))
            b.close(); eng.close()
    for i in (0, 1):
        n1, z1, g1, s1 = res[i]; n2, z2, g2, s2 = res[i + 2]
        if i == 0:   # EXACT: per-utterance scalars do not depend on the chunking at all
            assert np.array_equal(n1, n2) and np.array_equal(z1, z2)
        else:        # FAST (fused): scores do not depend on the chunking either, MFMA sums are per row
            np.testing.assert_allclose(n1, n2, rtol=1e-12, atol=1e-12)
            np.testing.assert_allclose(z1, z2, rtol=1e-13)
        np.testing.assert_allclose(g1, g2, rtol=1e-9, atol=1e-10 * np.abs(g1).max())
        np.testing.assert_allclose(s1, s2, rtol=1e-10)
    og, on, oz = Case(**kw).oracle_gradient()
    assert np.abs(res[2][2] - og).max() / np.abs(og).max() < 1e-9


def test_minibatch_reduce_and_optimizer_step():
    c = Case(L=4, D=3, in_w=2, Ts=[6, 5, 8], seed=33)
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
    orc.sgd_step(lam, acc, gsa, gg, 1.0, True, 1e-12)
    eng.sgd_step(1.0, True, 1e-12)
    assert np.array_equal(eng.get_lambda(), lam) and np.array_equal(eng.get_grad_sqr_acc(), gsa)
    assert np.array_equal(eng.get_lambda_acc(), acc)
    b.close(); eng.close()


def test_native_rccl_single_rank():
    c = Case(L=3, D=2, in_w=2, Ts=[5], seed=2)
    eng = c.engine(); b = c.batch(eng)
    eng.comm_init_single()
    eng.fb_batch(b); g = eng.get_grad()
    s = eng.allreduce_grad(active=True)
    assert s[3] == 1 and s[2] == 1 and np.array_equal(eng.get_grad(), g)
    b.close(); eng.close()


def test_error_behaviour():
    c = Case(L=3, D=2, in_w=2, Ts=[5], seed=2)
    eng = c.engine()
    bad = [np.array([scrf_amd.LAB_BAD, 3 * 2 + 1, scrf_amd.LAB_BAD, 0, 1], dtype=np.uint32)]  # label >= L*D
    b = eng.batch_from_frames(c.frames, bad, c.recipes)
    with pytest.raises(scrf_amd.ScrfError) as ei:
        eng.fb_batch(b)
    assert ei.value.code == 5
    b.close()
    with pytest.raises(scrf_amd.ScrfError) as ei:
        eng.batch_from_frames([np.zeros((0, 2), np.float32)], [np.zeros(0, np.uint32)], c.recipes)
    assert ei.value.code == 6  # "No features read from this sentence."
    with pytest.raises(scrf_amd.ScrfError):
        scrf_amd.Engine(scrf_amd.make_config(model_type=scrf_amd.STDFRAME, L=4, D=3, F=5))  # stdframe needs D=1
    eng.close()


def test_full_size_config2_utterances():
    """BASELINE config 2 at full size (L=48, D=25, 39-dim x 300 frames): oracle on 2 utterances,
    size-independent properties on all."""
    from scrf_amd import synth
    L, D, in_w, T, U = 48, 25, 39, 300, 6
    frames, labels, off = synth.make_batch(U, T, in_w, L, D)
    F = 8 * in_w + D
    lam = synth.make_lambda(L * (F + 1 + L))
    eng = scrf_amd.Engine(scrf_amd.make_config(L=L, D=D, F=F, precision=1)); eng.set_lambda(lam)
    fl = [frames[int(off[u]):int(off[u + 1])] for u in range(U)]
    ll = [labels[int(off[u]):int(off[u + 1])] for u in range(U)]
    b = eng.batch_from_frames(fl, ll)
    numer, zx = eng.fb_batch(b)
    g = eng.get_grad()
    assert np.isfinite(g).all() and (numer < zx).all()
    ocfg = orc.config(L=L, D=D, F=F); olay = orc.Layout(ocfg)
    og = np.zeros(olay.lambda_len)
    for u in range(2):
        X = orc.windows(fl[u], D)
        rc, og, on, oz = orc.seg_build_gradient(ocfg, olay, lam, X, ll[u], T, grad=og)
        assert rc == 0
        assert abs(on - numer[u]) <= 1e-10 * abs(on) and abs(oz - zx[u]) <= 1e-11 * abs(oz)
    eng.zero_grad()
    b2 = eng.batch_from_frames(fl[:2], ll[:2])
    eng.fb_batch(b2)
    assert np.abs(eng.get_grad() - og).max() / np.abs(og).max() < 1e-9
    # property: sum over labels of the state-bias gradient = (#true segments) - E[#segments]; and the
    # expected number of segment ends per utterance equals 1 at the last frame => bias grads are finite
    # and transition-bias gradient mass = state mass minus one segment per utterance
    nsf = F + 1
    sb = np.array([g[l * (nsf + L) + F] for l in range(L)]).sum()
    tb = sum(g[l * (nsf + L) + nsf:(l + 1) * (nsf + L)].sum() for l in range(L))
    assert abs((sb - tb)) < 1e-6 * U * T  # every segment but the last of an utterance has one outgoing transition
    labs, cost = eng.viterbi_batch(b)
    for u in range(U):
        durs = [int(x) // L + 1 for x in labs[u]]
        assert sum(durs) == T and max(durs) <= D
    So, Mo = orc.seg_scores(ocfg, olay, lam, orc.windows(fl[0], D), T)
    oa, ons, ofin = orc.seg_lattice_arcs(ocfg, So, Mo, T)
    ol, oc = orc.best_path(oa, ons, ofin)
    assert list(labs[0]) == list(ol) and np.float32(cost[0]).tobytes() == np.float32(oc).tobytes()
    ga, gns, gfin = eng.lattice_arcs(b, 0)
    assert ga.tobytes() == oa.tobytes()
    b.close(); b2.close(); eng.close()


def test_random_shape_sweep_through_the_fused_and_wavefront_kernels():
    """tools/fused_shape_sweep.py: 60 seeded random (D, W, L, T...) shapes -- W = 1 (division by
    magic number), L = 64 with large D (workgroup size chosen by the LDS fit), L just above 48 / 64,
    D = 2 .. 40, utterances shorter than D: FAST gradient with fused windows == materialised windows
    (1e-9) == oracle (1e-8, small cases); fast decode == EXACT decode bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fused_shape_sweep.py"), "60", "3"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") == 60


def test_several_wavefronts_per_sweep_kernel_through_the_same_sweep():
    """k_dp_lin_mv (SCRF_DPLIN_MV=1, opt-in: four wavefronts share one sweep's ring, durations and transition rows
    split between them) through 30 shapes of the same sweep, in its own process (the switch is read once)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRF_DPLIN_MV="1")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fused_shape_sweep.py"), "30", "7"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") == 30


@pytest.mark.parametrize("prec,split", [("1", "1"), ("3", "1"), ("3", "0")])
def test_posterior_walk_in_segments_and_in_one_piece(prec, split):
    """A launch of few utterances walks each utterance's posterior pass in several segments (k_post_z<., ., ., 1>: rings
    warmed from the stored vectors, boundary frames' sums added from both sides); large batches walk it in one piece.
    Both forms through 24 random shapes with a 260-frame utterance appended, FAST and FASTLIN, against the
    materialised-window kernels and the oracle (SCRF_POSTZ_SPLIT=0 forces the one-piece form, SCRF_DPLIN_MV=0 the
    one-wavefront recursion)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SWEEP_PREC=prec, SWEEP_LONG="1")
    if split == "0":
        env.update(SCRF_POSTZ_SPLIT="0", SCRF_DPLIN_MV="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fused_shape_sweep.py"), "24", "11"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert r.stdout.count("\nok ") + r.stdout.startswith("ok ") == 24


def test_one_wavefront_per_sweep_kernel_with_per_frame_matrices_at_small_batches():
    """With per-frame transition matrices the launcher takes k_dp_lin_mv<., 1> for launches of few sweeps -- every
    stdtrans case of this suite.  SCRF_DPLIN_MV=0 forces the single-wavefront k_dp_lin<., 1, .> (what large batches run)
    through the same 30 random cases."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCRF_DPLIN_MV="0")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "general_shape_sweep.py"), "30", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


@pytest.mark.parametrize("L,D,W", [(64, 25, 3), (64, 32, 2), (3, 5, 1)])
def test_regressions_found_by_the_shape_sweep(L, D, W, monkeypatch):
    """L = 64, D >= 17: twelve wavefronts' rings do not fit the LDS (the launch used to fail);
    W = 1: the expected-count kernel's magic-number division by W."""
    for prec in (0, 1):
        c = Case(L=L, D=D, in_w=W, Ts=[2, D, 2 * D + 3], seed=11, precision=prec, lam_scale=0.2)
        eng = c.engine(); b = c.batch(eng)
        numer, zx = eng.fb_batch(b)
        g = eng.get_grad()
        og, on, oz = c.oracle_gradient()
        assert np.abs(g - og).max() <= 1e-9 * np.abs(og).max()
        assert np.abs(zx - oz).max() <= 1e-11 * np.abs(oz).max()
        b.close(); eng.close()


def test_random_shape_sweep_through_every_path():
    """tools/general_shape_sweep.py: 60 seeded random cases over stdstate / stdtrans (context stream),
    EXACT / FAST / FAST32, tiny scratch budgets, L up to 300, D up to 45, W up to 90 -- gradient, Zx,
    numerator against the oracle within the precision's bound; Viterbi labels and cost bit-identical."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "general_shape_sweep.py"), "60", "7"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]


def test_batches_destroyed_with_kernels_in_flight_reuse_pooled_arrays_safely(monkeypatch):
    """A trainer makes one batch per minibatch and destroys it as soon as scrf_fb_batch has returned -- while the count
    kernels that read the batch's arrays are still running.  Batch arrays come from a pool whose blocks are handed out
    again only once the engine stream has passed their destroy call (round 4; before, every destroy synchronised the
    device).  Forty batches of alternating shapes back to back, gradients accumulated on the device: the sum must equal
    the sum of the same batches run one by one with the pool off."""
    shapes = [dict(L=48, D=10, in_w=13, Ts=[57, 33, 90, 12]), dict(L=48, D=10, in_w=13, Ts=[20, 64, 5]),
              dict(L=48, D=10, in_w=13, Ts=[130, 7, 41, 77, 19])]
    res = {}
    for pool in ("1", "0"):
        monkeypatch.setenv("SCRF_BATCH_POOL", pool)
        c0 = Case(seed=901, precision=scrf_amd.PREC_FAST, **shapes[0])
        eng = c0.engine()
        eng.zero_grad()
        for i in range(40):
            c = Case(seed=901 + i % 3, precision=scrf_amd.PREC_FAST, **shapes[i % 3])
            b = c.batch(eng)
            eng.fb_batch(b, want_scalars=False)   # returns once the recursion's status is known; contractions in flight
            b.close()
            if pool == "0" or i % 7 == 0:
                eng.synchronize()
        res[pool] = eng.get_grad().copy()
        eng.close()
    assert np.isfinite(res["1"]).all()
    assert np.abs(res["1"] - res["0"]).max() <= 1e-12 * np.abs(res["0"]).max()

"""ctypes/numpy front-end of the CPU oracle (oracle/libscrf_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libscrf_oracle.so")
REF_LOGMATH_PATH = os.path.join(ORACLE_DIR, "_ref", "libcrf_logmath_ref.so")

LAB_BAD = 0xFFFFFFFF
STDFRAME, STDSEG, STDSEG_NO_DUR, STDSEG_NO_DUR_NO_TRANSFTR, STDSEG_NO_DUR_NO_SEGTRANSFTR = range(5)


class OrcConfig(C.Structure):
    _fields_ = [
        ("model_type", C.c_uint32),
        ("num_labs", C.c_uint32),
        ("lab_max_dur", C.c_uint32),
        ("num_feas", C.c_uint32),
        ("use_state_ftrs", C.c_int32),
        ("state_fidx_start", C.c_uint32),
        ("state_fidx_end", C.c_uint32),
        ("use_trans_ftrs", C.c_int32),
        ("trans_fidx_start", C.c_uint32),
        ("trans_fidx_end", C.c_uint32),
        ("use_state_bias", C.c_int32),
        ("use_trans_bias", C.c_int32),
        ("state_bias_val", C.c_double),
        ("trans_bias_val", C.c_double),
        ("num_states", C.c_uint32),
    ]


class OrcLayout(C.Structure):
    _fields_ = [
        ("num_state_funcs", C.c_uint32),
        ("num_trans_funcs", C.c_uint32),
        ("lambda_len", C.c_uint32),
        ("state_idx", C.POINTER(C.c_uint32)),
        ("trans_idx", C.POINTER(C.c_uint32)),
    ]


ARC_DTYPE = np.dtype(
    [("src", "<i4"), ("ilabel", "<i4"), ("olabel", "<i4"), ("w", "<f4"), ("dst", "<i4")]
)

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        d = C.c_double
        _lib.orc_expE.restype = d
        _lib.orc_logE.restype = d
        _lib.orc_logadd2.restype = d
        _lib.orc_logadd_n.restype = d
        _lib.orc_logadd_max_n.restype = d
        _lib.orc_expE.argtypes = [d, C.POINTER(C.c_int)]
        _lib.orc_logE.argtypes = [d, C.POINTER(C.c_int)]
        _lib.orc_logadd2.argtypes = [d, d, C.POINTER(C.c_int)]
        _lib.orc_logadd_n.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        _lib.orc_logadd_max_n.argtypes = [C.c_void_p, d, C.c_int, C.POINTER(C.c_int)]
        _lib.orc_num_segs.restype = C.c_uint64
        _lib.orc_num_segs.argtypes = [C.c_uint32, C.c_uint32]
        _lib.orc_seg_base.restype = C.c_uint64
        _lib.orc_seg_base.argtypes = [C.c_uint32, C.c_uint32]
        _lib.orc_window_width.restype = C.c_uint32
        _lib.orc_seg_lattice_num_arcs.restype = C.c_uint64
        _lib.orc_seg_lattice_num_arcs_k.restype = C.c_uint64
        _lib.orc_seg_lattice_arcs.restype = C.c_uint64
        _lib.orc_segtrans_lattice_num_arcs.restype = C.c_uint64
        _lib.orc_segtrans_lattice_arcs.restype = C.c_uint64
        _lib.orc_nstate_lattice_num_arcs.restype = C.c_uint64
        _lib.orc_nstate_lattice_arcs.restype = C.c_uint64
        _lib.orc_stdseg_lattice_num_arcs.restype = C.c_uint64
        _lib.orc_stdseg_lattice_arcs.restype = C.c_uint64
        _lib.orc_frame_lattice_num_arcs.restype = C.c_uint64
        _lib.orc_frame_lattice_arcs.restype = C.c_uint64
        _lib.orc_best_path.restype = C.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def config(model_type=STDSEG_NO_DUR_NO_SEGTRANSFTR, L=3, D=3, F=4, sfs=0, sfe=None,
           use_trans_ftrs=False, tfs=0, tfe=None, use_state_ftrs=True, use_state_bias=True,
           use_trans_bias=True, state_bias_val=1.0, trans_bias_val=1.0, num_states=1):
    """Mirror of CRFTrain's set_fmap_config (CRFTrain/src/Main.cpp:372-430)."""
    if sfe is None or sfe < 0:
        sfe = F - 1
    if tfe is None or tfe < 0:
        tfe = F - 1
    return OrcConfig(model_type, L, D, F, int(use_state_ftrs), sfs, sfe, int(use_trans_ftrs), tfs,
                     tfe, int(use_state_bias), int(use_trans_bias), state_bias_val, trans_bias_val, num_states)


class Layout:
    def __init__(self, cfg):
        self.cfg = cfg
        self.c = OrcLayout()
        rc = lib().orc_layout_init(C.byref(cfg), C.byref(self.c))
        if rc != 0:
            raise ValueError("orc_layout_init failed: %d" % rc)
        L = cfg.num_labs
        self.num_state_funcs = self.c.num_state_funcs
        self.num_trans_funcs = self.c.num_trans_funcs
        self.lambda_len = self.c.lambda_len
        self.state_idx = np.ctypeslib.as_array(self.c.state_idx, (L,)).copy()
        self.trans_idx = np.ctypeslib.as_array(self.c.trans_idx, (L * L,)).copy()

    def __del__(self):
        try:
            lib().orc_layout_free(C.byref(self.c))
        except Exception:
            pass


def num_segs(T, D):
    return int(lib().orc_num_segs(T, D))


def seg_base(t, D):
    return int(lib().orc_seg_base(t, D))


def window_width(in_width, D, lctx=0, rctx=0, extract_seg=True):
    return int(lib().orc_window_width(C.c_uint32(in_width), C.c_uint32(D), C.c_uint32(lctx),
                                      C.c_uint32(rctx), C.c_int(int(extract_seg))))


def windows(frames, D, lctx=0, rctx=0, extract_seg=True, out=None, out_col=0):
    """frames: [T + lctx + rctx, in_width] float32 -> [N_seg, width] float32."""
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    T = frames.shape[0] - lctx - rctx
    W = frames.shape[1]
    width = window_width(W, D, lctx, rctx, extract_seg)
    if out is None:
        out = np.zeros((num_segs(T, D), width), dtype=np.float32)
    lib().orc_windows(_p(frames), C.c_uint32(T), C.c_uint32(W), C.c_uint32(D), C.c_uint32(lctx),
                      C.c_uint32(rctx), C.c_int(int(extract_seg)), _p(out),
                      C.c_uint32(out.shape[1]), C.c_uint32(out_col))
    return out


def group_labels(frame_labs, D, L):
    fl = np.ascontiguousarray(frame_labs, dtype=np.uint32)
    out = np.empty(fl.shape[0], dtype=np.uint32)
    lib().orc_group_labels(_p(fl), C.c_uint32(fl.shape[0]), C.c_uint32(D), C.c_uint32(L), _p(out))
    return out


def seg_scores(cfg, lay, lam, segftrs, T):
    L, D = cfg.num_labs, cfg.lab_max_dur
    segftrs = np.ascontiguousarray(segftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    S = np.zeros((num_segs(T, D), L), dtype=np.float64)
    M = np.zeros((T, L * L), dtype=np.float64)
    lib().orc_seg_scores(C.byref(cfg), C.byref(lay.c), _p(lam), _p(segftrs), C.c_uint32(T), _p(S), _p(M))
    return S, M


def seg_forward(cfg, S, M, T):
    L, D = cfg.num_labs, cfg.lab_max_dur
    ad = np.zeros((num_segs(T, D), L)); al = np.zeros((T, L)); apt = np.zeros((T, L))
    zx = C.c_double()
    rc = lib().orc_seg_forward(C.byref(cfg), _p(S), _p(M), C.c_uint32(T), _p(ad), _p(al), _p(apt), C.byref(zx))
    return rc, ad, al, apt, zx.value


def seg_backward(cfg, S, M, T):
    L = cfg.num_labs
    beta = np.zeros((T, L)); sd = np.zeros((T, L))
    rc = lib().orc_seg_backward(C.byref(cfg), _p(S), _p(M), C.c_uint32(T), _p(beta), _p(sd))
    return rc, beta, sd


def seg_posteriors(cfg, S, M, T):
    L, D = cfg.num_labs, cfg.lab_max_dur
    g = np.zeros((num_segs(T, D), L)); xi = np.zeros((T, L * L))
    zx = C.c_double()
    rc = lib().orc_seg_posteriors(C.byref(cfg), _p(S), _p(M), C.c_uint32(T), _p(g), _p(xi), C.byref(zx))
    return rc, g, xi, zx.value


def seg_build_gradient(cfg, lay, lam, segftrs, labels, T, grad=None):
    segftrs = np.ascontiguousarray(segftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    if grad is None:
        grad = np.zeros(lay.lambda_len, dtype=np.float64)
    numer = C.c_double(); zx = C.c_double()
    rc = lib().orc_seg_build_gradient(C.byref(cfg), C.byref(lay.c), _p(lam), _p(segftrs), _p(labels),
                                      C.c_uint32(T), _p(grad), C.byref(numer), C.byref(zx))
    return rc, grad, numer.value, zx.value


# ---- f3: STDSEG_NO_DUR (segment-dependent transition features) --------------------------- #
def segtrans_scores(cfg, lay, lam, segftrs, T):
    L, D = cfg.num_labs, cfg.lab_max_dur
    segftrs = np.ascontiguousarray(segftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    S = np.zeros((num_segs(T, D), L)); M2 = np.zeros((num_segs(T, D), L * L))
    lib().orc_segtrans_scores(C.byref(cfg), C.byref(lay.c), _p(lam), _p(segftrs), C.c_uint32(T), _p(S), _p(M2))
    return S, M2


def segtrans_forward(cfg, S, M2, T):
    L, D = cfg.num_labs, cfg.lab_max_dur
    ad = np.zeros((num_segs(T, D), L)); al = np.zeros((T, L))
    zx = C.c_double()
    rc = lib().orc_segtrans_forward(C.byref(cfg), _p(S), _p(M2), C.c_uint32(T), _p(ad), _p(al), C.byref(zx))
    return rc, ad, al, zx.value


def segtrans_backward(cfg, S, M2, T):
    beta = np.zeros((T, cfg.num_labs))
    rc = lib().orc_segtrans_backward(C.byref(cfg), _p(S), _p(M2), C.c_uint32(T), _p(beta))
    return rc, beta


def segtrans_posteriors(cfg, S, M2, T):
    L, D = cfg.num_labs, cfg.lab_max_dur
    g = np.zeros((num_segs(T, D), L)); xi = np.zeros((num_segs(T, D), L * L))
    zx = C.c_double()
    rc = lib().orc_segtrans_posteriors(C.byref(cfg), _p(S), _p(M2), C.c_uint32(T), _p(g), _p(xi), C.byref(zx))
    return rc, g, xi, zx.value


def segtrans_build_gradient(cfg, lay, lam, segftrs, labels, T, grad=None):
    segftrs = np.ascontiguousarray(segftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    if grad is None:
        grad = np.zeros(lay.lambda_len, dtype=np.float64)
    numer = C.c_double(); zx = C.c_double()
    rc = lib().orc_segtrans_build_gradient(C.byref(cfg), C.byref(lay.c), _p(lam), _p(segftrs), _p(labels),
                                           C.c_uint32(T), _p(grad), C.byref(numer), C.byref(zx))
    return rc, grad, numer.value, zx.value


def nstate_scores(cfg, lay, lam, ftrs, T):
    L, K = cfg.num_labs, cfg.num_states
    P = L // K
    ftrs = np.ascontiguousarray(ftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    S = np.zeros((T, L)); TD = np.zeros((T, L)); TO = np.zeros((T, L)); TE = np.zeros((T, P * P))
    lib().orc_nstate_scores(C.byref(cfg), C.byref(lay.c), _p(lam), _p(ftrs), C.c_uint32(T), _p(S), _p(TD), _p(TO), _p(TE))
    return S, TD, TO, TE


def nstate_forward(cfg, S, TD, TO, TE, T):
    al = np.zeros_like(S)
    zx = C.c_double()
    rc = lib().orc_nstate_forward(C.byref(cfg), _p(S), _p(TD), _p(TO), _p(TE), C.c_uint32(T), _p(al), C.byref(zx))
    return rc, al, zx.value


def nstate_backward(cfg, S, TD, TO, TE, T):
    be = np.zeros_like(S)
    rc = lib().orc_nstate_backward(C.byref(cfg), _p(S), _p(TD), _p(TO), _p(TE), C.c_uint32(T), _p(be))
    return rc, be


def nstate_build_gradient(cfg, lay, lam, ftrs, labels, T, grad=None):
    ftrs = np.ascontiguousarray(ftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    if grad is None:
        grad = np.zeros(lay.lambda_len, dtype=np.float64)
    numer = C.c_double(); zx = C.c_double()
    rc = lib().orc_nstate_build_gradient(C.byref(cfg), C.byref(lay.c), _p(lam), _p(ftrs), _p(labels), C.c_uint32(T), _p(grad),
                                         C.byref(numer), C.byref(zx))
    return rc, grad, numer.value, zx.value


def nstate_lattice_arcs(cfg, S, TD, TO, TE, T, norm=False, alpha_sum=0.0):
    n = int(lib().orc_nstate_lattice_num_arcs(C.c_uint32(T), C.c_uint32(cfg.num_labs), C.c_uint32(cfg.num_states)))
    arcs = np.zeros(n, dtype=ARC_DTYPE)
    ns = C.c_uint32(); fin = C.c_int32()
    na = lib().orc_nstate_lattice_arcs(C.byref(cfg), _p(S), _p(TD), _p(TO), _p(TE), C.c_uint32(T), C.c_int(int(norm)),
                                       C.c_double(alpha_sum), _p(arcs), C.byref(ns), C.byref(fin))
    assert na == n, (na, n)
    return arcs, ns.value, fin.value


def nstate_trans(cfg, TD, TO, TE, t, p, c):
    """transition score p -> c at frame t under the n-state topology, None when it is not allowed"""
    L, K = cfg.num_labs, cfg.num_states
    P = L // K
    if p == c:
        return TD[t, c]
    if c % K == 0:
        return TE[t, (p // K) * P + c // K] if (p + 1) % K == 0 else None
    return TO[t, c - 1] if p == c - 1 else None


def brute_force_nstate(cfg, S, TD, TO, TE, T):
    """every label sequence the n-state topology allows (any state may start and end an utterance, as
    computeFirstAlpha / computeAlphaSum have it): Zx, per-frame posteriors, the best sequence"""
    L = cfg.num_labs
    paths = []

    def rec(t, prev, score, seq):
        if t == T:
            paths.append((score, tuple(seq)))
            return
        for c in range(L):
            s = score + S[t, c]
            if prev is not None:
                tr = nstate_trans(cfg, TD, TO, TE, t, prev, c)
                if tr is None:
                    continue
                s = s + tr
            rec(t + 1, c, s, seq + [c])

    rec(0, None, 0.0, [])
    scores = np.array([p[0] for p in paths])
    mx = scores.max()
    Zx = mx + np.log(np.exp(scores - mx).sum())
    gamma = np.zeros((T, L))
    for sc, seq in paths:
        p = np.exp(sc - Zx)
        for t, c in enumerate(seq):
            gamma[t, c] += p
    best = max(range(len(paths)), key=lambda i: paths[i][0])
    return dict(Zx=Zx, gamma=gamma, n_paths=len(paths), paths=paths, best=paths[best])


def stdseg_scores(cfg, lay, lam, segftrs, T):
    NL, D = cfg.num_labs, cfg.lab_max_dur
    L = NL // D
    segftrs = np.ascontiguousarray(segftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    S = np.zeros((num_segs(T, D), L)); MX = np.zeros((num_segs(T, D), NL, L))
    lib().orc_stdseg_scores(C.byref(cfg), C.byref(lay.c), _p(lam), _p(segftrs), C.c_uint32(T), _p(S), _p(MX))
    return S, MX


def stdseg_forward(cfg, S, MX, T):
    al = np.zeros_like(S)
    zx = C.c_double()
    rc = lib().orc_stdseg_forward(C.byref(cfg), _p(S), _p(MX), C.c_uint32(T), _p(al), C.byref(zx))
    return rc, al, zx.value


def stdseg_backward(cfg, S, MX, T):
    beta = np.zeros_like(S)
    rc = lib().orc_stdseg_backward(C.byref(cfg), _p(S), _p(MX), C.c_uint32(T), _p(beta))
    return rc, beta


def stdseg_build_gradient(cfg, lay, lam, segftrs, labels, T, grad=None):
    segftrs = np.ascontiguousarray(segftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    if grad is None:
        grad = np.zeros(lay.lambda_len, dtype=np.float64)
    numer = C.c_double(); zx = C.c_double()
    rc = lib().orc_stdseg_build_gradient(C.byref(cfg), C.byref(lay.c), _p(lam), _p(segftrs), _p(labels),
                                         C.c_uint32(T), _p(grad), C.byref(numer), C.byref(zx))
    return rc, grad, numer.value, zx.value


def brute_force_stdseg(S, MX, T, L, D):
    """every (segmentation, labelling) of the STDSEG model: a segment (end, d, l) after a segment (., dp, p) scores
    S[(end,d)][l] + MX[(end,d)][(dp-1)*L + p][l] (the first segment of the utterance S alone).
    Returns Zx, gamma [N_seg, L], xi [N_seg, L*D, L], the paths and the best one."""
    paths = []

    def rec(t_next, prev, score, segs):
        if t_next == T:
            paths.append((score, tuple(segs)))
            return
        for d in range(1, D + 1):
            end = t_next + d - 1
            if end >= T:
                break
            row = seg_base(end, D) + d - 1
            for l in range(L):
                s = score + S[row, l]
                if prev is not None:
                    s = s + MX[row, prev, l]
                rec(end + 1, (d - 1) * L + l, s, segs + [(end, d, l)])

    rec(0, None, 0.0, [])
    scores = np.array([p[0] for p in paths])
    mx = scores.max()
    Zx = mx + np.log(np.exp(scores - mx).sum())
    gamma = np.zeros_like(S); xi = np.zeros_like(MX)
    for sc, segs in paths:
        p = np.exp(sc - Zx)
        for i, (end, d, l) in enumerate(segs):
            row = seg_base(end, D) + d - 1
            gamma[row, l] += p
            if i > 0:
                xi[row, (segs[i - 1][1] - 1) * L + segs[i - 1][2], l] += p
    best = max(range(len(paths)), key=lambda i: paths[i][0])
    return dict(Zx=Zx, gamma=gamma, xi=xi, n_paths=len(paths), paths=paths, best=paths[best])


def brute_force_segtrans(S, M2, T, L, D):
    """every (segmentation, labelling) of the STDSEG_NO_DUR model: a segment (end, d, l) after a segment
    labelled p scores S[(end,d)][l] + M2[(end,d)][p*L+l] (the first segment of the utterance S alone).
    Returns Zx, gamma [N_seg, L], xi [N_seg, L*L], the paths and the best one."""
    paths = []

    def rec(t_next, prev_lab, score, segs):
        if t_next == T:
            paths.append((score, tuple(segs)))
            return
        for d in range(1, D + 1):
            end = t_next + d - 1
            if end >= T:
                break
            row = seg_base(end, D) + d - 1
            for l in range(L):
                s = score + S[row, l]
                if prev_lab is not None:
                    s = s + M2[row, prev_lab * L + l]
                rec(end + 1, l, s, segs + [(end, d, l)])

    rec(0, None, 0.0, [])
    scores = np.array([p[0] for p in paths])
    mx = scores.max()
    Zx = mx + np.log(np.exp(scores - mx).sum())
    gamma = np.zeros_like(S); xi = np.zeros_like(M2)
    for sc, segs in paths:
        p = np.exp(sc - Zx)
        for i, (end, d, l) in enumerate(segs):
            row = seg_base(end, D) + d - 1
            gamma[row, l] += p
            if i > 0:
                xi[row, segs[i - 1][2] * L + l] += p
    best = max(range(len(paths)), key=lambda i: paths[i][0])
    return dict(Zx=Zx, gamma=gamma, xi=xi, n_paths=len(paths), paths=paths, best=paths[best])


def frame_build_gradient(cfg, lay, lam, ftrs, labels, T, grad=None):
    ftrs = np.ascontiguousarray(ftrs, dtype=np.float32)
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    if grad is None:
        grad = np.zeros(lay.lambda_len, dtype=np.float64)
    numer = C.c_double(); zx = C.c_double()
    rc = lib().orc_frame_build_gradient(C.byref(cfg), C.byref(lay.c), _p(lam), _p(ftrs), _p(labels),
                                        C.c_uint32(T), _p(grad), C.byref(numer), C.byref(zx))
    return rc, grad, numer.value, zx.value


def minibatch_reduce(sgrad, active):
    sgrad = np.ascontiguousarray(sgrad, dtype=np.float64)
    active = np.ascontiguousarray(active, dtype=np.int32)
    out = np.zeros(sgrad.shape[1])
    lib().orc_minibatch_reduce(_p(sgrad), C.c_uint32(sgrad.shape[0]), _p(active),
                               C.c_uint32(sgrad.shape[1]), _p(out))
    return out


def sgd_step(lam, lam_acc, gsa, grad, lr_or_eta, use_adagrad, eps=1e-12):
    lib().orc_sgd_step(_p(lam), _p(lam_acc), _p(gsa), _p(grad), C.c_uint32(lam.shape[0]),
                       C.c_double(lr_or_eta), C.c_int(int(use_adagrad)), C.c_double(eps))


def seg_lattice_arcs(cfg, S, M, T, norm=False, alpha_sum=0.0):
    L, D = cfg.num_labs, cfg.lab_max_dur
    n = int(lib().orc_seg_lattice_num_arcs_k(C.c_uint32(T), C.c_uint32(L), C.c_uint32(D), C.c_uint32(max(1, cfg.num_states))))
    arcs = np.zeros(n, dtype=ARC_DTYPE)
    ns = C.c_uint32(); fin = C.c_int32()
    na = lib().orc_seg_lattice_arcs(C.byref(cfg), _p(S), _p(M), C.c_uint32(T), C.c_int(int(norm)),
                                    C.c_double(alpha_sum), _p(arcs), C.byref(ns), C.byref(fin))
    assert na == n, (na, n)
    return arcs, ns.value, fin.value


def segtrans_lattice_arcs(cfg, S, M2, T, norm=False, alpha_sum=0.0):
    L, D = cfg.num_labs, cfg.lab_max_dur
    n = int(lib().orc_segtrans_lattice_num_arcs(C.c_uint32(T), C.c_uint32(L), C.c_uint32(D)))
    arcs = np.zeros(n, dtype=ARC_DTYPE)
    ns = C.c_uint32(); fin = C.c_int32()
    na = lib().orc_segtrans_lattice_arcs(C.byref(cfg), _p(S), _p(M2), C.c_uint32(T), C.c_int(int(norm)),
                                         C.c_double(alpha_sum), _p(arcs), C.byref(ns), C.byref(fin))
    assert na == n, (na, n)
    return arcs, ns.value, fin.value


def stdseg_lattice_arcs(cfg, S, MX, T, norm=False, alpha_sum=0.0):
    NL, D = cfg.num_labs, cfg.lab_max_dur
    n = int(lib().orc_stdseg_lattice_num_arcs(C.c_uint32(T), C.c_uint32(NL // D), C.c_uint32(D)))
    arcs = np.zeros(n, dtype=ARC_DTYPE)
    ns = C.c_uint32(); fin = C.c_int32()
    na = lib().orc_stdseg_lattice_arcs(C.byref(cfg), _p(S), _p(MX), C.c_uint32(T), C.c_int(int(norm)),
                                       C.c_double(alpha_sum), _p(arcs), C.byref(ns), C.byref(fin))
    assert na == n, (na, n)
    return arcs, ns.value, fin.value


def frame_lattice_arcs(cfg, S, M, T, norm=False, alpha_sum=0.0):
    L = cfg.num_labs
    n = int(lib().orc_frame_lattice_num_arcs(C.c_uint32(T), C.c_uint32(L)))
    arcs = np.zeros(n, dtype=ARC_DTYPE)
    ns = C.c_uint32(); fin = C.c_int32()
    na = lib().orc_frame_lattice_arcs(C.byref(cfg), _p(S), _p(M), C.c_uint32(T), C.c_int(int(norm)),
                                      C.c_double(alpha_sum), _p(arcs), C.byref(ns), C.byref(fin))
    assert na == n, (na, n)
    return arcs, ns.value, fin.value


def best_path(arcs, n_states, final_state, start=0):
    arcs = np.ascontiguousarray(arcs)
    out = np.zeros(max(1, n_states), dtype=np.uint32)
    cost = C.c_float()
    n = lib().orc_best_path(_p(arcs), C.c_uint64(arcs.shape[0]), C.c_uint32(n_states),
                            C.c_int32(start), C.c_int32(final_state), _p(out),
                            C.c_uint64(out.shape[0]), C.byref(cost))
    if n < 0:
        return None, None
    return out[:n].copy(), cost.value


def free_phone_decode(cfg, S, M, T):
    """CRFDecode's Viterbi decoder with its own free-phone-loop LM: segments (phone, dur, float
    weight, phone_start) first to last, and the best hypothesis weight."""
    ph = np.zeros(max(1, T), dtype=np.uint32); du = np.zeros(max(1, T), dtype=np.uint32)
    w = np.zeros(max(1, T), dtype=np.float32); ps = np.zeros(max(1, T), dtype=np.int32)
    n = C.c_uint32(); best = C.c_float()
    rc = lib().orc_free_phone_decode(C.byref(cfg), _p(S), _p(M), C.c_uint32(T), _p(ph), _p(du), _p(w), _p(ps),
                                     C.byref(n), C.byref(best))
    if rc != 0:
        return None, None
    k = n.value
    return list(zip(ph[:k].tolist(), du[:k].tolist(), w[:k].tolist(), ps[:k].tolist())), best.value


def bench_fb(cfg, lam, frames, labels, frame_off, in_width, n_threads):
    """Threaded CPU forward-backward over packed utterances; returns (rc, grad, numer, zx, seconds)."""
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.uint64)
    U = frame_off.shape[0] - 1
    lay = Layout(cfg)
    grad = np.zeros(lay.lambda_len); numer = np.zeros(U); zx = np.zeros(U)
    sec = C.c_double()
    rc = lib().orc_bench_fb(C.byref(cfg), _p(lam), _p(frames), _p(labels), _p(frame_off),
                            C.c_uint32(U), C.c_uint32(in_width), C.c_uint32(n_threads), _p(grad),
                            _p(numer), _p(zx), C.byref(sec))
    return rc, grad, numer, zx, sec.value


def bench_fb2(cfg, lam, frames, frames2, in_width2, ctx2, labels, frame_off, in_width, n_threads):
    """bench_fb with a second, context-padded stream (frames2: [sum_u (T_u + 2 ctx2)][in_width2]) joined behind the
    first stream's window columns: BASELINE config 3's input."""
    lam = np.ascontiguousarray(lam, dtype=np.float64)
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    frames2 = np.ascontiguousarray(frames2, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.uint32)
    frame_off = np.ascontiguousarray(frame_off, dtype=np.uint64)
    U = frame_off.shape[0] - 1
    lay = Layout(cfg)
    grad = np.zeros(lay.lambda_len); numer = np.zeros(U); zx = np.zeros(U)
    sec = C.c_double()
    rc = lib().orc_bench_fb2(C.byref(cfg), _p(lam), _p(frames), _p(frames2), C.c_uint32(in_width2), C.c_uint32(ctx2),
                             _p(labels), _p(frame_off), C.c_uint32(U), C.c_uint32(in_width), C.c_uint32(n_threads),
                             _p(grad), _p(numer), _p(zx), C.byref(sec))
    return rc, grad, numer, zx, sec.value


def bench_set_cpus(cpus):
    """pin the workers of the following bench_fb calls: worker s -> cpus[s % len(cpus)]; [] = floating threads"""
    arr = (C.c_int * max(1, len(cpus)))(*cpus)
    lib().orc_bench_set_cpus(arr, C.c_int(len(cpus)))


def bench_phases():
    """the reference's five phase timers (featLoad, transMat, alpha, beta, expF) of the last bench_fb call:
    microseconds summed over the worker threads"""
    out = (C.c_double * 5)()
    lib().orc_bench_phases(out)
    return dict(zip(("featLoad", "transMat", "alpha", "beta", "expF"), [float(v) for v in out]))


# --------------------------------------------------------------------------- #
# Independent brute-force enumeration of all labelled segmentations (tiny cases)
# --------------------------------------------------------------------------- #
def ns_allowed(K, p, c):
    """n-state topology (K states per phone): stay, advance to the next state, or end state -> any start state."""
    if K <= 1 or p == c:
        return True
    if c % K == 0:
        return (p + 1) % K == 0
    return p + 1 == c


def brute_force(S, M, T, L, D, K=1):
    """Enumerate every (segmentation, labelling); returns dict with Zx, gamma[N_seg,L],
    xi[T,L*L], best (score, labels as l+L*(d-1) per segment; ties -> first found).  K > 1: only the label
    sequences the n-state topology allows."""
    paths = []

    def rec(t_next, prev_lab, score, segs):
        if t_next == T:
            paths.append((score, tuple(segs)))
            return
        for d in range(1, D + 1):
            end = t_next + d - 1
            if end >= T:
                break
            row = seg_base(end, D) + d - 1
            for l in range(L):
                if prev_lab is not None and not ns_allowed(K, prev_lab, l):
                    continue
                s = score + S[row, l]
                if prev_lab is not None:
                    s = s + M[t_next, prev_lab * L + l]
                rec(end + 1, l, s, segs + [(end, d, l)])

    rec(0, None, 0.0, [])
    scores = np.array([p[0] for p in paths])
    mx = scores.max()
    Zx = mx + np.log(np.exp(scores - mx).sum())
    gamma = np.zeros_like(S)
    xi = np.zeros((T, L * L))
    for sc, segs in paths:
        p = np.exp(sc - Zx)
        for i, (end, d, l) in enumerate(segs):
            gamma[seg_base(end, D) + d - 1, l] += p
            if i + 1 < len(segs):
                xi[end, l * L + segs[i + 1][2]] += p
    best = max(range(len(paths)), key=lambda i: paths[i][0])
    return dict(Zx=Zx, gamma=gamma, xi=xi, n_paths=len(paths), paths=paths, best=paths[best])

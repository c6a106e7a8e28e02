import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(os.path.dirname(HERE))  # asr-craft_amd/

LAB_BAD = 0xFFFFFFFF
STDFRAME, STDSEG, STDSEG_NO_DUR, STDSEG_NO_DUR_NO_TRANSFTR, STDSEG_NO_DUR_NO_SEGTRANSFTR = range(5)
STDSTATE, STDTRANS = 0, 1
PREC_EXACT, PREC_FAST, PREC_FAST32, PREC_FASTLIN = 0, 1, 2, 3
ABI_VERSION = 1
MAX_STREAMS = 3
N_PHASES = 10
PHASES = ("windows", "scores", "fwd_bwd", "expf", "reduce", "viterbi", "total", "k_scores", "k_dp", "k_expf")

ARC_DTYPE = np.dtype([("src", "<i4"), ("ilabel", "<i4"), ("olabel", "<i4"), ("w", "<f4"), ("dst", "<i4")])


class ScrfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("scrf error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("model_type", C.c_uint32), ("map_type", C.c_uint32),
        ("num_labs", C.c_uint32), ("num_feas", C.c_uint32), ("num_states", C.c_uint32),
        ("lab_max_dur", C.c_uint32), ("use_state_ftrs", C.c_int32),
        ("state_fidx_start", C.c_uint32), ("state_fidx_end", C.c_uint32),
        ("use_trans_ftrs", C.c_int32), ("trans_fidx_start", C.c_uint32), ("trans_fidx_end", C.c_uint32),
        ("use_state_bias", C.c_int32), ("use_trans_bias", C.c_int32),
        ("state_bias_val", C.c_double), ("trans_bias_val", C.c_double),
        ("device_id", C.c_int32), ("train_precision", C.c_uint32), ("scratch_bytes", C.c_uint64),
    ]


class StreamRecipe(C.Structure):
    _fields_ = [("in_width", C.c_uint32), ("left_ctx", C.c_uint32), ("right_ctx", C.c_uint32),
                ("extract_seg_ftr", C.c_int32)]


class Utt(C.Structure):
    _fields_ = [("T", C.c_uint32), ("windows", C.c_void_p), ("frames", C.c_void_p * MAX_STREAMS),
                ("labels", C.c_void_p)]


def lib_path():
    # SCRF_AMD_LIB: another build of the same library (kernel A/B measurements); the default is the in-tree build
    return os.environ.get("SCRF_AMD_LIB") or os.path.join(PKG_ROOT, "lib", "libscrf_amd.so")


_lib = None


def load_library():
    """Loads libscrf_amd.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise ScrfError(-1, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
        _lib = C.CDLL(p)
        _lib.scrf_last_error.restype = C.c_char_p
        _lib.scrf_last_error.argtypes = [C.c_void_p]
    return _lib


def window_width(in_width, D, lctx=0, rctx=0, extract_seg=True):
    """io/CRF_InFtrStream_SeqMultiWindow.cpp:50-125"""
    if D == 1:
        return (lctx + 1 + rctx) * in_width
    if extract_seg:
        return 8 * in_width + D + (lctx + rctx) * in_width
    return (lctx + 1 + rctx) * in_width


def make_config(model_type=STDSEG_NO_DUR_NO_SEGTRANSFTR, L=48, D=25, F=337, sfs=0, sfe=-1, use_trans_ftrs=False,
                tfs=0, tfe=-1, use_state_ftrs=True, use_state_bias=True, use_trans_bias=True,
                state_bias_val=1.0, trans_bias_val=1.0, device_id=0, precision=PREC_EXACT, scratch_bytes=0, num_states=1):
    """Same meaning as CRFTrain's set_fmap_config (CRFTrain/src/Main.cpp:372-430)."""
    if sfe is None or sfe < 0:
        sfe = F - 1
    if tfe is None or tfe < 0:
        tfe = F - 1
    return Config(ABI_VERSION, model_type, STDTRANS if use_trans_ftrs else STDSTATE, L, F, num_states, D,
                  int(use_state_ftrs), sfs, sfe, int(use_trans_ftrs), tfs, tfe, int(use_state_bias),
                  int(use_trans_bias), state_bias_val, trans_bias_val, device_id, precision, scratch_bytes)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Batch:
    def __init__(self, eng, handle, n, keep):
        self.eng, self.handle, self.n = eng, handle, n
        self._keep = keep
        nu = C.c_uint32(); nf = C.c_uint64(); ns = C.c_uint64(); na = C.c_uint64()
        eng._chk(eng.lib.scrf_batch_info(eng.h, handle, C.byref(nu), C.byref(nf), C.byref(ns), C.byref(na)))
        self.n_frames, self.n_segs, self.n_arcs = nf.value, ns.value, na.value

    def close(self):
        if self.handle:
            self.eng.lib.scrf_batch_destroy(self.eng.h, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RaggedLabels:
    """Read-only sequence of per-utterance label arrays over one flat array + offsets (scrf_viterbi_batch's output)."""
    __slots__ = ("flat", "off")

    def __init__(self, flat, off):
        self.flat, self.off = flat, off

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, u):
        if isinstance(u, slice):
            return [self[i] for i in range(*u.indices(len(self)))]
        if u < 0:
            u += len(self)
        if not 0 <= u < len(self):
            raise IndexError(u)
        return self.flat[int(self.off[u]):int(self.off[u + 1])]

    def __iter__(self):
        for u in range(len(self)):
            yield self[u]


class Engine:
    def __init__(self, cfg):
        self.lib = load_library()
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.lib.scrf_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            raise ScrfError(rc, self.lib.scrf_last_error(None).decode())
        n = C.c_uint32()
        self._chk(self.lib.scrf_lambda_len(self.h, C.byref(n)))
        self.lambda_len = n.value
        self.L, self.D, self.F = cfg.num_labs, cfg.lab_max_dur, cfg.num_feas

    def _chk(self, rc):
        if rc != 0:
            raise ScrfError(rc, self.lib.scrf_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.lib.scrf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- layout / model state
    def state_idx(self, clab, fno=0):
        v = C.c_uint32(); self._chk(self.lib.scrf_state_idx(self.h, clab, fno, C.byref(v))); return v.value

    def trans_idx(self, plab, clab, fno=0):
        v = C.c_uint32(); self._chk(self.lib.scrf_trans_idx(self.h, plab, clab, fno, C.byref(v))); return v.value

    def num_state_funcs(self):
        v = C.c_uint32(); self._chk(self.lib.scrf_num_state_funcs(self.h, C.byref(v))); return v.value

    def num_trans_funcs(self):
        v = C.c_uint32(); self._chk(self.lib.scrf_num_trans_funcs(self.h, C.byref(v))); return v.value

    def set_lambda(self, lam):
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        self._chk(self.lib.scrf_set_lambda(self.h, _p(lam), C.c_uint32(lam.shape[0])))

    def _set(self, fn, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        self._chk(fn(self.h, _p(v), C.c_uint32(v.shape[0])))

    def set_lambda_acc(self, v): self._set(self.lib.scrf_set_lambda_acc, v)
    def set_grad_sqr_acc(self, v): self._set(self.lib.scrf_set_grad_sqr_acc, v)

    def _get(self, fn):
        out = np.empty(self.lambda_len, dtype=np.float64)
        self._chk(fn(self.h, _p(out), C.c_uint32(self.lambda_len)))
        return out

    def get_lambda(self): return self._get(self.lib.scrf_get_lambda)
    def get_lambda_acc(self): return self._get(self.lib.scrf_get_lambda_acc)
    def get_grad_sqr_acc(self): return self._get(self.lib.scrf_get_grad_sqr_acc)
    def get_grad(self): return self._get(self.lib.scrf_get_grad)

    def zero_grad(self): self._chk(self.lib.scrf_zero_grad(self.h))
    def synchronize(self): self._chk(self.lib.scrf_synchronize(self.h))

    def set_stream(self, stream_ptr):
        self._chk(self.lib.scrf_set_stream(self.h, C.c_void_p(stream_ptr)))

    def set_grad_buffer(self, dptr):
        self._chk(self.lib.scrf_set_grad_buffer(self.h, C.c_void_p(dptr)))

    def grad_device_ptr(self):
        p = C.c_void_p(); self._chk(self.lib.scrf_grad_device_ptr(self.h, C.byref(p))); return p.value

    def add_grad(self, g):
        g = np.ascontiguousarray(g, dtype=np.float64)
        self._chk(self.lib.scrf_add_grad(self.h, _p(g), C.c_uint32(g.shape[0])))

    def batch_sums(self):
        s = np.zeros(3); self._chk(self.lib.scrf_get_batch_sums(self.h, _p(s))); return s

    # ---- batches
    def batch_from_frames(self, frames_list, labels_list=None, recipes=None, streams2=None):
        """frames_list: list of [T(+ctx), in_width] arrays (stream 0); streams2: optional list of
        lists for the further streams; recipes: list of StreamRecipe (default: one segment-feature stream)."""
        n = len(frames_list)
        all_streams = [frames_list] + (streams2 or [])
        if recipes is None:
            recipes = [StreamRecipe(frames_list[0].shape[1], 0, 0, 1)]
        assert len(recipes) == len(all_streams)
        keep = []
        utts = (Utt * n)()
        for u in range(n):
            r0 = recipes[0]
            utts[u].T = all_streams[0][u].shape[0] - r0.left_ctx - r0.right_ctx
            utts[u].windows = None
            for s, st in enumerate(all_streams):
                a = np.ascontiguousarray(st[u], dtype=np.float32)
                keep.append(a)
                utts[u].frames[s] = a.ctypes.data
            if labels_list is not None:
                lb = np.ascontiguousarray(labels_list[u], dtype=np.uint32)
                keep.append(lb)
                utts[u].labels = lb.ctypes.data
        rec = (StreamRecipe * len(recipes))(*recipes)
        hb = C.c_void_p()
        self._chk(self.lib.scrf_batch_create(self.h, utts, C.c_uint32(n), C.c_uint32(len(recipes)), rec, C.byref(hb)))
        return Batch(self, hb, n, None)

    def batch_from_windows(self, windows_list, T_list, labels_list=None):
        n = len(windows_list)
        keep = []
        utts = (Utt * n)()
        for u in range(n):
            a = np.ascontiguousarray(windows_list[u], dtype=np.float32)
            keep.append(a)
            utts[u].T = int(T_list[u])
            utts[u].windows = a.ctypes.data
            if labels_list is not None:
                lb = np.ascontiguousarray(labels_list[u], dtype=np.uint32)
                keep.append(lb)
                utts[u].labels = lb.ctypes.data
        hb = C.c_void_p()
        self._chk(self.lib.scrf_batch_create(self.h, utts, C.c_uint32(n), C.c_uint32(0), None, C.byref(hb)))
        return Batch(self, hb, n, None)

    # ---- hot path
    def fb_batch(self, batch, want_scalars=True):
        if want_scalars:
            numer = np.zeros(batch.n); zx = np.zeros(batch.n)
            self._chk(self.lib.scrf_fb_batch(self.h, batch.handle, _p(numer), _p(zx)))
            return numer, zx
        self._chk(self.lib.scrf_fb_batch(self.h, batch.handle, None, None))
        return None, None

    def num_segs(self, T):
        D = self.D
        return T * (T + 1) // 2 if T < D else D * (D + 1) // 2 + (T - D) * D

    def scores(self, batch, u, T):
        """S [N_seg, L]; M [T, L*L] -- for STDSEG_NO_DUR one transition matrix per window: [N_seg, L*L]; for STDSEG
        (L = all labels, La = L / D phones) S [N_seg, La] and M [N_seg, L, La] (previous FULL label x phone)"""
        if self.cfg.num_states > 1 and self.cfg.model_type == STDFRAME:   # n-state frame model: M [T, 2 L + P*P] = self | c -> c+1 | end of p -> start of q
            P = self.L // self.cfg.num_states
            S = np.zeros((T, self.L)); M = np.zeros((T, 2 * self.L + P * P))
            self._chk(self.lib.scrf_scores(self.h, batch.handle, C.c_uint32(u), _p(S), _p(M)))
            return S, M
        if self.cfg.model_type == STDSEG:
            La = self.L // self.D
            S = np.zeros((self.num_segs(T), La)); M = np.zeros((self.num_segs(T), self.L, La))
            self._chk(self.lib.scrf_scores(self.h, batch.handle, C.c_uint32(u), _p(S), _p(M)))
            return S, M
        S = np.zeros((self.num_segs(T), self.L))
        M = np.zeros((self.num_segs(T) if self.cfg.model_type == STDSEG_NO_DUR else T, self.L * self.L))
        self._chk(self.lib.scrf_scores(self.h, batch.handle, C.c_uint32(u), _p(S), _p(M)))
        return S, M

    def windows(self, batch, u, T):
        X = np.zeros((self.num_segs(T), self.F), dtype=np.float32)
        self._chk(self.lib.scrf_windows(self.h, batch.handle, C.c_uint32(u), _p(X)))
        return X

    def forward_backward(self, batch, u, T, prec=PREC_EXACT):
        if self.cfg.num_states > 1 and self.cfg.model_type == STDFRAME:
            al = np.zeros((T, self.L)); be = np.zeros((T, self.L))
            zx = C.c_double()
            self._chk(self.lib.scrf_forward_backward(self.h, batch.handle, C.c_uint32(u), C.c_uint32(prec), None, _p(al), _p(be), C.byref(zx)))
            return None, al, be, zx.value
        if self.cfg.model_type == STDSEG:   # the nodes' alpha / beta over full labels: [N_seg, La] each
            La = self.L // self.D
            al = np.zeros((self.num_segs(T), La)); be = np.zeros((self.num_segs(T), La))
            zx = C.c_double()
            self._chk(self.lib.scrf_forward_backward(self.h, batch.handle, C.c_uint32(u), C.c_uint32(prec), _p(al), None, _p(be), C.byref(zx)))
            return al, None, be, zx.value
        ad = np.zeros((self.num_segs(T), self.L)); al = np.zeros((T, self.L)); be = np.zeros((T, self.L))
        zx = C.c_double()
        self._chk(self.lib.scrf_forward_backward(self.h, batch.handle, C.c_uint32(u), C.c_uint32(prec), _p(ad), _p(al),
                                                 _p(be), C.byref(zx)))
        return ad, al, be, zx.value

    # ---- decode
    def lattice_arcs(self, batch, u, norm=False):
        na = C.c_uint64(); ns = C.c_uint32(); fin = C.c_int32()
        self._chk(self.lib.scrf_lattice_arcs(self.h, batch.handle, C.c_uint32(u), C.c_int(int(norm)), None,
                                             C.byref(na), C.byref(ns), C.byref(fin)))
        arcs = np.zeros(na.value, dtype=ARC_DTYPE)
        self._chk(self.lib.scrf_lattice_arcs(self.h, batch.handle, C.c_uint32(u), C.c_int(int(norm)), _p(arcs),
                                             C.byref(na), C.byref(ns), C.byref(fin)))
        return arcs, ns.value, fin.value

    def viterbi_batch(self, batch):
        """Best-path segment labels of every utterance and the path costs.  The labels come back as the C ABI delivers
        them -- one flat array and U + 1 offsets -- behind a sequence view (`RaggedLabels`): `labs[u]` is utterance u's
        array; nothing is copied per utterance (a Python loop over 4096 utterances cost a quarter of the call)."""
        cap = batch.n_frames
        labs = np.empty(cap, dtype=np.uint32); off = np.empty(batch.n + 1, dtype=np.uint64)
        cost = np.empty(batch.n, dtype=np.float32)
        self._chk(self.lib.scrf_viterbi_batch(self.h, batch.handle, _p(labs), C.c_uint64(cap), _p(off), _p(cost)))
        return RaggedLabels(labs, off), cost

    def batch_is_fused(self, batch):
        f = C.c_int()
        self._chk(self.lib.scrf_batch_is_fused(self.h, batch.handle, C.byref(f)))
        return bool(f.value)

    def batch_fused_mode(self, batch):
        """0: materialised windows; 1: fused window synthesis; 2: fused with the linear window average (FASTLIN)."""
        f = C.c_int()
        self._chk(self.lib.scrf_batch_is_fused(self.h, batch.handle, C.byref(f)))
        return f.value

    def set_frame_mass_check(self, on=True):
        """posterior-mass self-checks with the frame model's bounds on a segmental engine (scrf_set_frame_mass_check)"""
        self._chk(self.lib.scrf_set_frame_mass_check(self.h, C.c_int(int(on))))

    def decode_stats(self):
        """(arc weights recomputed in reference order, chunks sent back to the EXACT path) since create."""
        a = C.c_uint64(); b = C.c_uint64()
        self._chk(self.lib.scrf_decode_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- reduce + optimizer
    def allreduce_grad(self, active=True):
        s = np.zeros(4)
        self._chk(self.lib.scrf_allreduce_grad(self.h, C.c_int(int(active)), _p(s)))
        return s

    def comm_init_single(self):
        """RCCL communicator of one rank (exercises the native path on a single GPU)."""
        uid = (C.c_char * 128)()
        rc = self.lib.scrf_comm_unique_id(uid)
        if rc != 0:
            raise ScrfError(rc, self.lib.scrf_last_error(None).decode())
        self._chk(self.lib.scrf_comm_init(self.h, uid, 0, 1))

    def comm_unique_id(self):
        """128-byte RCCL id (rank 0 calls it and hands the bytes to the other ranks)"""
        uid = (C.c_char * 128)()
        rc = self.lib.scrf_comm_unique_id(uid)
        if rc != 0:
            raise ScrfError(rc, self.lib.scrf_last_error(None).decode())
        return bytes(uid.raw)

    def comm_init(self, uid, rank, world):
        """collective: every rank of the communicator calls it with rank 0's id"""
        buf = (C.c_char * 128).from_buffer_copy(uid)
        self._chk(self.lib.scrf_comm_init(self.h, buf, C.c_int(rank), C.c_int(world)))

    def scale_grad(self, s): self._chk(self.lib.scrf_scale_grad(self.h, C.c_double(s)))

    def sgd_step(self, lr_or_eta, use_adagrad=False, eps=1e-12):
        self._chk(self.lib.scrf_sgd_step(self.h, C.c_double(lr_or_eta), C.c_int(int(use_adagrad)), C.c_double(eps)))

    # ---- measurement
    def enable_timing(self, on=True): self._chk(self.lib.scrf_enable_timing(self.h, C.c_int(int(on))))

    def kernel_timing(self):
        """[(kernel name, total ms, launches)] of the last timed fb_batch / viterbi_batch, in launch order."""
        buf = C.create_string_buffer(1 << 16)
        self._chk(self.lib.scrf_kernel_timing(self.h, buf, C.c_size_t(len(buf))))
        out = []
        for line in buf.value.decode().splitlines():
            name, ms, n = line.split("\t")
            out.append((name, float(ms), int(n)))
        return out

    def train_stats(self):
        """batches redone through the log-domain recursion after the linear-domain one raised NUMERIC"""
        v = C.c_uint64()
        self._chk(self.lib.scrf_train_stats(self.h, C.byref(v)))
        return v.value

    def last_timing(self):
        ms = (C.c_float * N_PHASES)(); nl = (C.c_uint32 * N_PHASES)()
        self._chk(self.lib.scrf_last_timing(self.h, ms, nl))
        return {PHASES[i]: (ms[i], nl[i]) for i in range(N_PHASES)}

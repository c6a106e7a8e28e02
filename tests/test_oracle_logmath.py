"""a1: the oracle's LogMath restatement is PINNED bit-for-bit to the reference's own
CRF_LogMath (utils/CRF_LogMath.cpp:41-224): (1) against golden vectors produced by the
reference build (tests/golden/gen_logmath_golden.py), (2) live against
oracle/_ref/libcrf_logmath_ref.so when that build is present (this container)."""
import ctypes as C
import os

import numpy as np
import pytest

import orc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "logmath_ref.npz"))


def _same(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_constants():
    lib = orc.lib()
    log0 = C.c_double.in_dll(lib, "ORC_LOG0").value
    assert log0 == G["LOG0"][0] == -np.finfo(np.float64).max


def test_expE_logE_golden():
    lib = orc.lib()
    for name, fn in (("exp", lib.orc_expE), ("log", lib.orc_logE)):
        xin, xout, thr = G[name + "_in"], G[name + "_out"], G[name + "_threw"]
        for x, y, t in zip(xin, xout, thr):
            e = C.c_int(0)
            v = fn(float(x), C.byref(e))
            assert (e.value != 0) == bool(t), (name, x)
            if not t:
                assert _same(v, y), (name, x, v, y)


def test_logadd2_golden():
    lib = orc.lib()
    for a, b, y, t in zip(G["add2_a"], G["add2_b"], G["add2_out"], G["add2_threw"]):
        e = C.c_int(0)
        v = lib.orc_logadd2(float(a), float(b), C.byref(e))
        assert (e.value != 0) == bool(t), (a, b)
        if not t:
            assert _same(v, y), (a, b, v, y)


def test_logadd_vec_golden():
    lib = orc.lib()
    for R, y, t, ym, tm in zip(G["vec_in"], G["vec_out"], G["vec_threw"], G["vecmax_out"], G["vecmax_threw"]):
        R = np.ascontiguousarray(R[~np.isnan(R)])
        e = C.c_int(0)
        v = lib.orc_logadd_n(R.ctypes.data, len(R), C.byref(e))
        assert (e.value != 0) == bool(t)
        if not t:
            assert _same(v, y)
        e = C.c_int(0)
        v = lib.orc_logadd_max_n(R.ctypes.data, float(R.max()), len(R), C.byref(e))
        assert (e.value != 0) == bool(tm)
        if not tm:
            assert _same(v, ym)


@pytest.mark.skipif(not os.path.exists(orc.REF_LOGMATH_PATH), reason="reference build absent (GPU box)")
def test_live_against_reference_build():
    ref = C.CDLL(orc.REF_LOGMATH_PATH)
    ref.ref_logadd_n.restype = C.c_double
    ref.ref_logadd_n.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib = orc.lib()
    rng = np.random.RandomState(7)
    for _ in range(2000):
        n = int(rng.randint(1, 60))
        R = np.ascontiguousarray(rng.normal(rng.normal(0, 300), 10 ** rng.uniform(-3, 2.5), n))
        t = C.c_int(0); e = C.c_int(0)
        a = ref.ref_logadd_n(R.ctypes.data, n, C.byref(t))
        b = lib.orc_logadd_n(R.ctypes.data, n, C.byref(e))
        assert bool(t.value) == (e.value != 0)
        if not t.value:
            assert _same(a, b)

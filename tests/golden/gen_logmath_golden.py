"""Generates tests/golden/logmath_ref.npz by calling the REFERENCE's own CRF_LogMath
(compiled from /root/reference into oracle/_ref/libcrf_logmath_ref.so by oracle/Makefile).
Run in the build container only (the reference does not travel):
    make -C oracle && python tests/golden/gen_logmath_golden.py
The .npz holds inputs and the reference's outputs (data only)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ref = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcrf_logmath_ref.so"))
for f in ("ref_LOG0", "ref_LN_MAX", "ref_expE", "ref_logE", "ref_logadd2", "ref_logadd_n", "ref_logadd_max_n"):
    getattr(ref, f).restype = C.c_double
ref.ref_expE.argtypes = [C.c_double, C.POINTER(C.c_int)]
ref.ref_logE.argtypes = [C.c_double, C.POINTER(C.c_int)]
ref.ref_logadd2.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_int)]
ref.ref_logadd_n.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
ref.ref_logadd_max_n.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int)]

rng = np.random.RandomState(20261003)
out = {"LOG0": np.array([ref.ref_LOG0()]), "LN_MAX": np.array([ref.ref_LN_MAX()])}

# scalar exp/log incl. the throwing edges
xs = np.concatenate([rng.uniform(-745, 709.7, 400), rng.normal(0, 5, 400),
                     [0.0, -0.0, 709.782712893384, 709.78271289338397, 709.7827128933841, 710.0,
                      -1e308, 1e308, np.inf, -np.inf, np.nan, -745.2, -800.0]])
e_val = np.zeros_like(xs); e_thr = np.zeros(xs.shape, np.int32)
for i, x in enumerate(xs):
    t = C.c_int(0); e_val[i] = ref.ref_expE(x, C.byref(t)); e_thr[i] = t.value
ls = np.concatenate([np.exp(rng.uniform(-700, 700, 400)), rng.uniform(0, 4, 200),
                     [0.0, -0.0, 1.0, 5e-324, 1.7976931348623157e308, -1.0, np.inf, np.nan]])
l_val = np.zeros_like(ls); l_thr = np.zeros(ls.shape, np.int32)
for i, x in enumerate(ls):
    t = C.c_int(0); l_val[i] = ref.ref_logE(x, C.byref(t)); l_thr[i] = t.value
out.update(exp_in=xs, exp_out=e_val, exp_threw=e_thr, log_in=ls, log_out=l_val, log_threw=l_thr)

# pairwise logAdd
a = np.concatenate([rng.normal(0, 50, 500), [0, 1e300, -1e300, ref.ref_LOG0(), 3.0]])
b = np.concatenate([rng.normal(0, 50, 500), [0, 1e300, 1e300, ref.ref_LOG0(), ref.ref_LOG0()]])
p_val = np.zeros_like(a); p_thr = np.zeros(a.shape, np.int32)
for i in range(len(a)):
    t = C.c_int(0); p_val[i] = ref.ref_logadd2(a[i], b[i], C.byref(t)); p_thr[i] = t.value
out.update(add2_a=a, add2_b=b, add2_out=p_val, add2_threw=p_thr)

# vector logAdd, lengths like L=48 / D<=25 / 200, various spreads
vecs, v_out, v_thr, vm_out, vm_thr = [], [], [], [], []
for n in (1, 2, 3, 10, 25, 48, 200):
    for spread in (0.01, 1.0, 30.0, 400.0):
        for _ in range(6):
            R = np.ascontiguousarray(rng.normal(-100 * rng.rand(), spread, n))
            t = C.c_int(0); v = ref.ref_logadd_n(R.ctypes.data, n, C.byref(t))
            t2 = C.c_int(0); vm = ref.ref_logadd_max_n(R.ctypes.data, float(R.max()), n, C.byref(t2))
            vecs.append(np.pad(R, (0, 200 - n), constant_values=np.nan)); v_out.append(v); v_thr.append(t.value)
            vm_out.append(vm); vm_thr.append(t2.value)
# LOG0 members (the reference's exp underflows to 0 for them)
R = np.array([ref.ref_LOG0(), -3.0, ref.ref_LOG0(), -2.5]); t = C.c_int(0)
vecs.append(np.pad(R, (0, 196), constant_values=np.nan)); v_out.append(ref.ref_logadd_n(R.ctypes.data, 4, C.byref(t))); v_thr.append(t.value)
t2 = C.c_int(0); vm_out.append(ref.ref_logadd_max_n(R.ctypes.data, -2.5, 4, C.byref(t2))); vm_thr.append(t2.value)
out.update(vec_in=np.array(vecs), vec_out=np.array(v_out), vec_threw=np.array(v_thr, np.int32),
           vecmax_out=np.array(vm_out), vecmax_threw=np.array(vm_thr, np.int32))
np.savez_compressed(os.path.join(HERE, "logmath_ref.npz"), **out)
print("wrote logmath_ref.npz:", {k: v.shape for k, v in out.items()})

"""Phase times inside k_scores_fused / k_expf_fused from a -DFU_PROF=1 build of the library (debug only):
   make -C asr-craft_amd/csrc OUTDIR=$PWD/asr-craft_amd/lib_prof EXTRA=-DFU_PROF=1
   SCRF_AMD_LIB=$PWD/asr-craft_amd/lib_prof/libscrf_amd.so python tools/fused_phases.py
Wave 0 of every workgroup stamps the phase boundaries (s_memtime, 100 MHz ticks) and the kernel sums the differences."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "asr-craft_amd", "python"))
import torch  # noqa
import scrf_amd
from scrf_amd import synth

L, D, IN_W, T, U = 48, 25, 39, 300, 4096
frames, labels, off = synth.make_batch(U, T, IN_W, L, D, seed=1234)
eng = scrf_amd.Engine(scrf_amd.make_config(L=L, D=D, F=8 * IN_W + D, device_id=0, scratch_bytes=96 << 30, precision=int(os.environ.get("PREC", "3"))))
eng.set_lambda(synth.make_lambda(eng.lambda_len))
batch = eng.batch_from_frames([frames[int(off[u]):int(off[u + 1])] for u in range(U)],
                              [labels[int(off[u]):int(off[u + 1])] for u in range(U)])
lib = scrf_amd.load_library()
buf = (C.c_ulonglong * 16)()
for rep in range(3):
    eng.zero_grad(); eng.fb_batch(batch, want_scalars=False); eng.synchronize()
    lib.scrf_debug_fused_prof(buf, 1)
v = list(buf)
n = max(1, v[15])
print("k_scores_fused: tiles", v[15], " (s_memtime ticks per tile, summed over wave 0 of every workgroup)")
for i, name in enumerate(["stage(rest)", "scan", "mfma", "pstage", "epilogue", "st:desc", "st:frames", "st:recs"]):
    if v[i]:
        print("  %-12s %10.1f" % (name, v[i] / n))
print("  %-12s %10.1f" % ("sum", sum(v[:8]) / n))
n = max(1, v[14])
print("k_expf_fused: tiles", v[14])
for i, name in [(8, "stage | producers: build"), (9, "scan | producers: barrier wait"), (10, "mfma | consumers: mfma"), (11, "consumers: barrier wait")]:
    print("  %-32s %10.1f" % (name, v[i] / n))

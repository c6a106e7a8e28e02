"""scrf_amd -- thin ctypes binding of libscrf_amd.so (include/scrf_abi.h).

Plumbing for tests and bench.py only: the product is the C-ABI library built from
asr-craft_amd/csrc (HIP kernels for gfx950) and the C++ adaptor classes in asr-craft_amd/host.
There is no CPU fallback: loading fails loudly if the library is missing, and engine
creation fails with SCRF_ERR_NO_DEVICE without a GPU.
"""
from .engine import (LAB_BAD, STDFRAME, STDSEG, STDSEG_NO_DUR, STDSEG_NO_DUR_NO_SEGTRANSFTR, STDSEG_NO_DUR_NO_TRANSFTR,  # noqa: F401
                     STDSTATE, STDTRANS, PREC_EXACT, PREC_FAST, PREC_FAST32, PREC_FASTLIN, ARC_DTYPE, Batch, Engine, ScrfError, StreamRecipe, lib_path,
                     load_library, make_config, window_width)

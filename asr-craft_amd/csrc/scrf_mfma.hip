// scrf_mfma.hip -- MFMA (v_mfma_f64_16x16x4_f64) contractions of the FAST training path.
#include "scrf_kernels.h"

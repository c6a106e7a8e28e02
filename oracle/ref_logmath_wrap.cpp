/*
 * ref_logmath_wrap.cpp -- C-linkage shim over the REFERENCE's own CRF_LogMath, so tests
 * can call it through ctypes.  TEST INFRASTRUCTURE ONLY.
 *
 * This file contains no reference code: it includes the reference header from where it
 * lies (/root/reference/CRF/src/utils/CRF_LogMath.h, via -I in oracle/Makefile) and is
 * linked with CRF_LogMath.cpp compiled from that same place into oracle/_ref/.
 * CRF_LogMath.{h,cpp} + CRF_Utils.h are the only reference sources that build without
 * QuickNet3/OpenFST/cblas.
 */
#include "CRF_LogMath.h"

extern "C" {

double ref_LOG0(void) { return CRF_LogMath::LOG0; }
double ref_LN_MAX(void) { return CRF_LogMath::CRF_DBL_LN_MAX; }

/* *threw is set to 1 when the reference throws (overflow_error / runtime_error) */
double ref_expE(double a, int* threw) {
  try { return CRF_LogMath::expE(a); } catch (std::exception&) { *threw = 1; return 0.0; }
}
double ref_logE(double a, int* threw) {
  try { return CRF_LogMath::logE(a); } catch (std::exception&) { *threw = 1; return 0.0; }
}
double ref_logadd2(double a, double b, int* threw) {
  try { return CRF_LogMath::logAdd(a, b); } catch (std::exception&) { *threw = 1; return 0.0; }
}
double ref_logadd_n(double* R, int n, int* threw) {
  try { return CRF_LogMath::logAdd(R, n); } catch (std::exception&) { *threw = 1; return 0.0; }
}
double ref_logadd_max_n(double* R, double max, int n, int* threw) {
  try { return CRF_LogMath::logAdd(R, max, n); } catch (std::exception&) { *threw = 1; return 0.0; }
}

}

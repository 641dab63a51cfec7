// version.hip -- build identification for libvcvits_hip.so
#include "common.h"
extern "C" const char* vcv_version(void) {
  return "vcvits_hip gfx950 fp32-mfma(32x32x2) conv family; built " __DATE__ " " __TIME__;
}

// Deterministic mode (VCVITS_DETERMINISTIC=1 / vcv_set_deterministic): launchers that normally split a reduction over
// several workgroups and combine with fp32 atomics (bias / thin / grouped / register-staged weight gradients, the
// one-output-channel forward) run it unsplit -- one writer per output element, a fixed summation order -- and the MFMA
// weight-gradient kernels combine through slabs (VcvWgradArgs.slab).  Slower; results are bit-reproducible run to run.
static int g_det = [] { const char* e = getenv("VCVITS_DETERMINISTIC"); return e && e[0] == '1' ? 1 : 0; }();
extern "C" int vcv_set_deterministic(int on) {
  g_det = on ? 1 : 0;
  return VCV_OK;
}
extern "C" int vcv_get_deterministic(void) { return g_det; }

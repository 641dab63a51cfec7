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

// Device-side seed offset for the counter-based dropout masks: when set (non-null), the forward dropout / attention
// launchers pass the pointer to their kernels, which add *ptr to the host-supplied seed.  A launch sequence captured in a
// HIP graph bakes the host seed into its kernel arguments; bumping the device word between replays gives every replay
// fresh masks (vcvits_amd/light/graphed.py).  A backward kernel that regenerates a mask reads the same pointer: inside a
// captured training pass forward and backward see the same value; eager passes run with the pointer unset.
static const unsigned long long* g_seed_off = nullptr;
extern "C" int vcv_set_seed_offset_ptr(const void* dev_ptr) {
  g_seed_off = (const unsigned long long*)dev_ptr;
  return VCV_OK;
}
extern "C" const void* vcv_get_seed_offset_ptr(void) { return g_seed_off; }

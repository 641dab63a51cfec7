// version.hip -- build identification for libvcvits_hip.so
#include "common.h"
#include <cstring>
extern "C" const char* vcv_version(void) {
  return "vcvits_hip gfx950 fp32-mfma(32x32x2) conv family; built " __DATE__ " " __TIME__;
}

// Deterministic mode (VCVITS_DETERMINISTIC=1 / vcv_set_deterministic): launchers that normally split a reduction over
// several workgroups and combine with fp32 atomics (bias / thin / grouped / register-staged weight gradients, the
// one-output-channel forward) run it unsplit -- one writer per output element, a fixed summation order -- and the MFMA
// weight-gradient kernels combine through slabs (VcvWgradArgs.slab).  Slower; results are bit-reproducible run to run.
extern "C" int vcv_set_deterministic(int on) {
  vcv_tuning().deterministic = on ? 1 : 0;
  return VCV_OK;
}
extern "C" int vcv_get_deterministic(void) { return vcv_tuning().deterministic; }

// ---- the tuning table (tuning.h) ------------------------------------------------------------------------------------------
namespace {
struct TuningKey { const char* name; int VcvTuning::*field; };
#define TK(f) {#f, &VcvTuning::f}
const TuningKey kTuningKeys[] = {
    TK(xcd_remap), TK(pk_ws), TK(pk_ws_bf16), TK(pk_x4), TK(pk_vec), TK(x3_variant), TK(x3_v6), TK(x3_js2), TK(x3_old_ks), TK(x3_all),
    TK(x3_terms), TK(wgrad_dma), TK(wgrad_tile), TK(wgrad_verbose), TK(wgrad_bf16_ws), TK(wgrad_finish_vec), TK(bias_rows),
    TK(c1_chunk), TK(m1_lds), TK(c1_wgrad_pairs), TK(thin_wgrad_wgs), TK(act_grad_vec), TK(ln_regs), TK(stft_wave), TK(attn_rows),
    TK(zero_memset), TK(pack_tile), TK(pack_tile_bf16), TK(pair_dbg), TK(pair_grid), TK(pair_stream), TK(deterministic)};
#undef TK
int VcvTuning::*find_key(const char* key, size_t len) {
  for (const TuningKey& k : kTuningKeys)
    if (strlen(k.name) == len && strncmp(k.name, key, len) == 0) return k.field;
  return nullptr;
}
VcvTuning make_tuning() {
  VcvTuning t;
  const char* det = getenv("VCVITS_DETERMINISTIC");
  if (det && det[0] == '1') t.deterministic = 1;
  // VCVITS_TUNING="key=value,key=value": unknown keys are reported once on stderr and ignored
  const char* e = getenv("VCVITS_TUNING");
  while (e && *e) {
    const char* end = strchr(e, ',');
    const size_t n = end ? (size_t)(end - e) : strlen(e);
    const char* eq = (const char*)memchr(e, '=', n);
    if (eq) {
      int VcvTuning::*f = find_key(e, (size_t)(eq - e));
      if (f) t.*f = atoi(eq + 1);
      else fprintf(stderr, "vcvits_hip: VCVITS_TUNING: unknown key '%.*s'\n", (int)(eq - e), e);
    } else if (n) {
      fprintf(stderr, "vcvits_hip: VCVITS_TUNING: '%.*s' is not key=value\n", (int)n, e);
    }
    e = end ? end + 1 : nullptr;
  }
  if (t.x3_terms != 9) t.x3_terms = 6;
  return t;
}
}  // namespace
VcvTuning& vcv_tuning() {
  static VcvTuning t = make_tuning();
  return t;
}
extern "C" int vcv_tuning_set(const char* key, int value) {
  int VcvTuning::*f = key ? find_key(key, strlen(key)) : nullptr;
  if (!f) return VCV_EINVAL;
  vcv_tuning().*f = value;
  return VCV_OK;
}
extern "C" int vcv_tuning_get(const char* key, int* value) {
  int VcvTuning::*f = key ? find_key(key, strlen(key)) : nullptr;
  if (!f || !value) return VCV_EINVAL;
  *value = vcv_tuning().*f;
  return VCV_OK;
}

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

// common.h -- shared device helpers for the gfx950 kernels of libvcvits_hip.so
#pragma once
#include "tuning.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "vcvits_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define VCV_LDS_LIMIT (160 * 1024)

__device__ __forceinline__ float vcv_leaky(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float vcv_dleaky(float aux, float s) { return aux > 0.f ? 1.f : s; }

// operand transform applied while staging global -> LDS
__device__ __forceinline__ float vcv_tf(float v, int tf, const float* aux, size_t idx, float slope) {
  if (tf == VCV_TF_LEAKY) return vcv_leaky(v, slope);
  if (tf == VCV_TF_DLEAKY) return v * vcv_dleaky(aux[idx], slope);
  if (tf == VCV_TF_DRELU) return aux[idx] > 0.f ? v : 0.f;
  if (tf == VCV_TF_DTANH) { const float a = aux[idx]; return v * (1.f - a * a); }
  if (tf == VCV_TF_DLOGCLAMP) { const float a = aux[idx]; return a > logf(slope) ? v * expf(-a) : 0.f; }
  return v;
}

// the same with the aux element already in a register
__device__ __forceinline__ float vcv_tf_val(float v, int tf, float a, float slope) {
  if (tf == VCV_TF_LEAKY) return vcv_leaky(v, slope);
  if (tf == VCV_TF_DLEAKY) return v * vcv_dleaky(a, slope);
  if (tf == VCV_TF_DRELU) return a > 0.f ? v : 0.f;
  if (tf == VCV_TF_DTANH) return v * (1.f - a * a);
  if (tf == VCV_TF_DLOGCLAMP) return a > logf(slope) ? v * expf(-a) : 0.f;
  return v;
}

__device__ __forceinline__ float vcv_act(float v, int act, float slope) {
  switch (act) {
    case VCV_ACT_LEAKY: return vcv_leaky(v, slope);
    case VCV_ACT_RELU: return v > 0.f ? v : 0.f;
    case VCV_ACT_TANH: return tanhf(v);
    case VCV_ACT_LOGCLAMP: return logf(fmaxf(v, slope));
    default: return v;
  }
}

static inline int vcv_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VCV_OK : VCV_EHIP;
}

static inline int vcv_cdiv(int a, int b) { return (a + b - 1) / b; }

// Zero `bytes` (a multiple of 4) at p on `stream` with a KERNEL.  The launchers used hipMemsetAsync; inside a stream capture
// that records a memset NODE, and tuning key zero_memset = 1 keeps it (A/B: light/graphed.py's replays).
static __global__ void vcv_zero_words_kernel(uint32_t* __restrict__ p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = 0u;
}
static inline hipError_t vcv_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (vcv_tuning().zero_memset || (bytes & 3) || ((uintptr_t)p & 3)) return hipMemsetAsync(p, 0, bytes, st);
  const size_t n = bytes / 4;
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(vcv_zero_words_kernel, dim3(blocks), dim3(256), 0, st, (uint32_t*)p, n);
  return hipGetLastError();
}

extern "C" int vcv_get_deterministic(void);  // version.hip
extern "C" const void* vcv_get_seed_offset_ptr(void);  // version.hip

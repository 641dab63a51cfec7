// probe: fp32 MFMA issue rate vs waves per SIMD, accumulators per wave, and an LDS-read pattern like conv_dma's loop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ACC, int LDSREAD>
__global__ void __launch_bounds__(1024) probe(float* out, int iters, const float* src) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = src[i & 1023];
  __syncthreads();
  f32x16 acc[ACC];
  for (int a = 0; a < ACC; ++a) for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float av = tid * 0.001f, bv = 1.0f;
  const float* lp = lds + (tid & 63);
  for (int it = 0; it < iters; ++it) {
    if (LDSREAD) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a0 = lp[(j * 192 + it * 8) & 8191 & ~63 | 0], a1 = lp[((j * 192 + 64 + it * 8) & 8191 & ~63)], b0 = lp[((j * 192 + 128 + it * 8) & 8191 & ~63)];
#pragma unroll
        for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32((a & 1) ? a1 : a0, b0, acc[a], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int a = 0; a < ACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < ACC; ++a) for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int ACC, int LDSREAD>
void run(int waves_per_simd, float* out, const float* src) {
  const int threads = 64 * 4 * waves_per_simd;
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<ACC, LDSREAD>), dim3(256), dim3(threads), 32768, 0, out, 10, src);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<ACC, LDSREAD>), dim3(256), dim3(threads), 32768, 0, out, iters, src);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * (threads / 64) * iters * 8.0 * ACC * (32.0 * 32 * 2 * 2);
  printf("waves/SIMD %d  acc %d  lds %d : %.1f TFLOP/s  (%.3f ms)\n", waves_per_simd, ACC, LDSREAD, flops / ms / 1e9, ms);
}

int main() {
  float *out, *src;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&src, 4096);
  hipMemset(src, 0, 4096);
  for (int w = 1; w <= 4; ++w) { run<1, 0>(w, out, src); run<2, 0>(w, out, src); run<4, 0>(w, out, src); }
  for (int w = 1; w <= 4; ++w) { run<2, 1>(w, out, src); run<4, 1>(w, out, src); }
  return 0;
}

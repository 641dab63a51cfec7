// probe: (1) does ds_read_b128 at 4-byte alignment return the right data on gfx950, (2) wgrad-like loop rate:
// per 8 reduction steps a wave reads TM aligned + TN (mis)aligned b128 fragments and issues 4*TM*TN fp32 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline f32x4 lds_read128(const float* p) {
  f32x4 v;
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)p;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a));
  return v;
}

__global__ void check(float* out) {
  __shared__ float lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (float)i;
  __syncthreads();
  f32x4 v = lds_read128(lds + threadIdx.x * 5 + 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}

template <int TM, int TN, int MODE>  // MODE 0: B aligned, 1: per-lane shift (n%5)*P4 P4=3 floats, 2: all lanes +1 float
__global__ void __launch_bounds__(1024) loop(float* out, int iters, int pitchA, int pitchB) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 24576; i += blockDim.x) lds[i] = 0.001f * (i & 255);
  __syncthreads();
  const int l31 = lane & 31, h = lane >> 5;
  f32x16 acc[TM][TN];
  for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const float* Ab = lds + ((wave & 1) * TM * 32 + l31) * pitchA + 4 * h;
  int sh = MODE == 0 ? 0 : MODE == 1 ? ((l31 + wave) % 5) * 3 : 1;
  const float* Bb = lds + 12288 + ((l31 / 5) + (wave >> 1) * 7) * pitchB + sh + 4 * h;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {  // 8 positions each
      f32x4 a[TM], b[TN];
#pragma unroll
      for (int t = 0; t < TM; ++t) a[t] = lds_read128(Ab + t * 32 * pitchA + j * 8);
#pragma unroll
      for (int t = 0; t < TN; ++t) b[t] = lds_read128(Bb + t * 7 * pitchB + j * 8);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
          for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][e], b[u][e], acc[t][u], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int TM, int TN, int MODE>
void run(int waves, float* out, int pA, int pB) {
  const int threads = 64 * waves, iters = 1000;
  hipFuncSetAttribute((const void*)loop<TM, TN, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((loop<TM, TN, MODE>), dim3(256), dim3(threads), 98304, 0, out, 10, pA, pB);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((loop<TM, TN, MODE>), dim3(256), dim3(threads), 98304, 0, out, iters, pA, pB);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = 256.0 * waves * iters * 8.0 * 4 * TM * TN * (32.0 * 32 * 2 * 2);
  printf("waves %2d  tile %dx%d  mode %d pitch %d/%d : %.1f TFLOP/s\n", waves, TM, TN, MODE, pA, pB, flops / ms / 1e9);
}

int main() {
  float *out, *h = (float*)malloc(1024);
  hipMalloc(&out, 256 * 1024 * 4);
  hipLaunchKernelGGL(check, dim3(1), dim3(64), 0, 0, out);
  hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) for (int j = 0; j < 4; ++j) if (h[i * 4 + j] != (float)(i * 5 + 1 + j)) ++bad;
  printf("misaligned ds_read_b128: %d bad of 256 (lane1: %g %g %g %g)\n", bad, h[4], h[5], h[6], h[7]);
  for (int w = 4; w <= 16; w *= 2) {
    run<2, 2, 0>(w, out, 68, 84); run<2, 2, 1>(w, out, 68, 84); run<2, 2, 2>(w, out, 68, 84);
    run<2, 2, 1>(w, out, 68, 85); run<2, 2, 1>(w, out, 68, 81);
    run<2, 1, 0>(w, out, 68, 84); run<2, 1, 1>(w, out, 68, 84);
    run<1, 2, 1>(w, out, 68, 84);
  }
  return 0;
}

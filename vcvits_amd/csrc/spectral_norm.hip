// spectral_norm.hip -- torch.nn.utils.spectral_norm (dim 0, one power iteration) for the discriminators'
// use_spectral_norm=True branch (reference: vits/model/discriminators/discriminator.py:17,52 pick it as norm_f;
// multi_scale_discriminator.py:13-19).  The weight is a row-major [R, N] matrix (R = output channels, N = the rest);
// u [R] and v [N] are the layer's persistent power-iteration vectors.
//
//   training forward:  v <- normalize(W^T u);  u <- normalize(W v);  sigma = u . (W v);  W_sn = W / sigma
//   eval forward:      sigma = u . (W v) with the stored u, v
//   backward:          dW = (dW_sn - <dW_sn, W_sn> u v^T) / sigma          (u, v are constants of the graph, as in torch)
//
// HBM-bound vector work on weights of at most 1024 x 5120 floats (21 MB): W is read three times in a training forward
// (column sums, row dots, scale), all coalesced along N; every reduction has one writer per output and a fixed order
// (no atomics), so the result is the same run to run.
#include "common.h"

namespace {

constexpr int SN_COLS = 64;   // columns per workgroup of the W^T u pass
constexpr int SN_RG = 4;      // row groups (of 64 lanes each) sharing those columns
constexpr int SN_PART = 256;  // partial sums of the backward inner product

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

// t[j] = sum_r w[r][j] * u[r]: a wave's lanes = 64 consecutive columns (coalesced rows), four waves split the rows
__global__ __launch_bounds__(256) void sn_wt_u_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                     float* __restrict__ t, int R, int N) {
  __shared__ float part[SN_RG][SN_COLS];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * SN_COLS + lane;
  float acc = 0.f;
  if (j < N) {
    int r = rg;
    for (; r + 3 * SN_RG < R; r += 4 * SN_RG) {  // four loads in flight
      const float a0 = w[(size_t)r * N + j], a1 = w[(size_t)(r + SN_RG) * N + j];
      const float a2 = w[(size_t)(r + 2 * SN_RG) * N + j], a3 = w[(size_t)(r + 3 * SN_RG) * N + j];
      acc += a0 * u[r];
      acc += a1 * u[r + SN_RG];
      acc += a2 * u[r + 2 * SN_RG];
      acc += a3 * u[r + 3 * SN_RG];
    }
    for (; r < R; r += SN_RG) acc += w[(size_t)r * N + j] * u[r];
  }
  part[rg][lane] = acc;
  __syncthreads();
  if (rg == 0 && j < N) t[j] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// out = x / max(||x||, eps)  (F.normalize); one workgroup.  With `s_for_sigma` (the W v pass of the same forward):
// sigma = out . x as well.
__global__ __launch_bounds__(1024) void sn_normalize_kernel(const float* __restrict__ x, float* __restrict__ out, int n,
                                                            float eps, float* __restrict__ sigma) {
  __shared__ float red[16];
  float ss = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) ss += x[i] * x[i];
  const float nrm = sqrtf(block_sum(ss, red));
  const float inv = 1.f / fmaxf(nrm, eps);
  float dot = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float o = x[i] * inv;
    out[i] = o;
    dot += o * x[i];
  }
  if (sigma) {
    const float d = block_sum(dot, red);
    if (threadIdx.x == 0) *sigma = d;
  }
}

// s[r] = sum_j w[r][j] * v[j]: one workgroup per row
__global__ __launch_bounds__(256) void sn_w_v_kernel(const float* __restrict__ w, const float* __restrict__ v,
                                                    float* __restrict__ s, int N) {
  __shared__ float red[4];
  const float* row = w + (size_t)blockIdx.x * N;
  float acc = 0.f;
  for (int j = threadIdx.x; j < N; j += 256) acc += row[j] * v[j];
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) s[blockIdx.x] = t;
}

// sigma = u . s with the stored u (eval mode: no power iteration)
__global__ __launch_bounds__(1024) void sn_dot_kernel(const float* __restrict__ u, const float* __restrict__ s, int n,
                                                      float* __restrict__ sigma) {
  __shared__ float red[16];
  float d = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) d += u[i] * s[i];
  const float t = block_sum(d, red);
  if (threadIdx.x == 0) *sigma = t;
}

__global__ void sn_scale_kernel(const float* __restrict__ w, const float* __restrict__ sigma, float* __restrict__ out,
                                size_t n) {
  const float inv = 1.f / *sigma;  // torch divides (weight / sigma); the reciprocal differs by at most one rounding
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = w[i] * inv;
}

// partial[b] = sum over the workgroup's stripe of dw_sn * w_sn (fixed stripes: reproducible)
__global__ __launch_bounds__(256) void sn_inner_kernel(const float* __restrict__ dwsn, const float* __restrict__ wsn,
                                                      float* __restrict__ partial, size_t n) {
  __shared__ float red[4];
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)SN_PART * 256) acc += dwsn[i] * wsn[i];
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// dw[r][j] = (dw_sn[r][j] - inner * u[r] * v[j]) / sigma; one workgroup per (row, 1024-column chunk)
__global__ __launch_bounds__(256) void sn_bwd_kernel(const float* __restrict__ dwsn, const float* __restrict__ u,
                                                    const float* __restrict__ v, const float* __restrict__ sigma,
                                                    const float* __restrict__ partial, float* __restrict__ dw, int N) {
  __shared__ float red[4];
  const float inner = block_sum(partial[threadIdx.x], red);  // SN_PART == blockDim.x, same order in every workgroup
  const float inv = 1.f / *sigma;
  const int r = blockIdx.x;
  const float iu = inner * u[r];
  const size_t base = (size_t)r * N;
  for (int j = blockIdx.y * 1024 + threadIdx.x; j < min(N, (int)(blockIdx.y + 1) * 1024); j += 256)
    dw[base + j] = (dwsn[base + j] - iu * v[j]) * inv;
}

}  // namespace

#define ST ((hipStream_t)stream)

// work: R + N floats of scratch
extern "C" int vcv_spectral_norm_fwd(const float* w, float* u, float* v, float* w_sn, float* sigma, float* work, int R, int N,
                                     int power_iteration, float eps, void* stream) {
  if (!w || !u || !v || !w_sn || !sigma || !work || R <= 0 || N <= 0 || !(eps > 0.f)) return VCV_EINVAL;
  float* s = work;      // [R]  W v
  float* t = work + R;  // [N]  W^T u
  if (power_iteration) {
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3((N + SN_COLS - 1) / SN_COLS), dim3(256), 0, ST, w, (const float*)u, t, R, N);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, ST, (const float*)t, v, N, eps, (float*)nullptr);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(R), dim3(256), 0, ST, w, (const float*)v, s, N);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, ST, (const float*)s, u, R, eps, sigma);
  } else {
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(R), dim3(256), 0, ST, w, (const float*)v, s, N);
    hipLaunchKernelGGL(sn_dot_kernel, dim3(1), dim3(1024), 0, ST, (const float*)u, (const float*)s, R, sigma);
  }
  const size_t n = (size_t)R * N;
  const unsigned blocks = (unsigned)((n + 1023) / 1024 < 2048 ? (n + 1023) / 1024 : 2048);
  hipLaunchKernelGGL(sn_scale_kernel, dim3(blocks), dim3(256), 0, ST, w, (const float*)sigma, w_sn, n);
  return vcv_check_launch();
}

// work: 256 floats of scratch.  u, v, sigma: the values the forward divided by (copies taken after its power iteration).
extern "C" int vcv_spectral_norm_bwd(const float* dw_sn, const float* w_sn, const float* u, const float* v, const float* sigma,
                                     float* dw, float* work, int R, int N, void* stream) {
  if (!dw_sn || !w_sn || !u || !v || !sigma || !dw || !work || R <= 0 || N <= 0) return VCV_EINVAL;
  const size_t n = (size_t)R * N;
  hipLaunchKernelGGL(sn_inner_kernel, dim3(SN_PART), dim3(256), 0, ST, dw_sn, w_sn, work, n);
  hipLaunchKernelGGL(sn_bwd_kernel, dim3(R, (N + 1023) / 1024), dim3(256), 0, ST, dw_sn, u, v, sigma, (const float*)work, dw, N);
  return vcv_check_launch();
}

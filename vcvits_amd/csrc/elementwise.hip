// elementwise.hip -- HBM-bound helpers of the vcvits hot path (weight-norm reparametrisation,
// reflect pad / average pool of the discriminator inputs, loss reductions, AdamW).
// All are streaming kernels: coalesced reads of [B, C, T] rows, one pass, wavefront (64-lane)
// shuffles for the reductions.
#include "common.h"
#include <stdlib.h>

namespace {

__device__ __forceinline__ float wave_sum(float s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  return s;
}

// block-wide sum for 256-thread blocks; result valid in every thread
__device__ __forceinline__ float block_sum256(float s, float* red) {
  s = wave_sum(s);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---- weight norm (torch.nn.utils.weight_norm, dim=0): w = g * v / ||v||_row -----------------
__global__ void __launch_bounds__(256)
weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ w,
                       float* __restrict__ norm, int C) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* vr = v + (size_t)r * C;
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) { const float a = vr[i]; s += a * a; }
  s = block_sum256(s, red);
  const float nrm = sqrtf(s);
  const float sc = g[r] / nrm;
  float* wr = w + (size_t)r * C;
  for (int i = threadIdx.x; i < C; i += 256) wr[i] = vr[i] * sc;
  if (threadIdx.x == 0) norm[r] = nrm;
}

__global__ void __launch_bounds__(256)
weight_norm_bwd_kernel(const float* __restrict__ dw, const float* __restrict__ v,
                       const float* __restrict__ g, const float* __restrict__ norm,
                       float* __restrict__ dv, float* __restrict__ dg, int C) {
  __shared__ float red[4];
  const int r = blockIdx.x;
  const float* vr = v + (size_t)r * C;
  const float* dwr = dw + (size_t)r * C;
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) s += dwr[i] * vr[i];
  s = block_sum256(s, red);
  const float nrm = norm[r], gg = g[r];
  const float a = gg / nrm, bcoef = gg * s / (nrm * nrm * nrm);
  float* dvr = dv + (size_t)r * C;
  for (int i = threadIdx.x; i < C; i += 256) dvr[i] = a * dwr[i] - bcoef * vr[i];
  if (threadIdx.x == 0) dg[r] = s / nrm;
}

// ---- batched weight norm: one launch for every weight-normed layer of a module ------------------------
// items[i] = {v ptr, g ptr, w offset (floats, into wbuf), first row (into the row-indexed norm buffer), R, C,
//             dw ptr, dv ptr, dg ptr, accumulate (backward: dv / dg are written, or added onto when set)};
// one workgroup per row of every tensor.
struct WnItem { long long v, g, w_off, row0, R, C, dw, dv, dg, acc; };

__device__ __forceinline__ int wn_find(const WnItem* __restrict__ items, int n, int row) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].row0 <= row) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
weight_norm_many_fwd_kernel(const WnItem* __restrict__ items, int n, float* __restrict__ wbuf, float* __restrict__ norm) {
  __shared__ float red[4];
  const WnItem it = items[wn_find(items, n, blockIdx.x)];
  const int r = blockIdx.x - (int)it.row0, C = (int)it.C;
  const float* vr = (const float*)it.v + (size_t)r * C;
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) { const float a = vr[i]; s += a * a; }
  s = block_sum256(s, red);
  const float nrm = sqrtf(s);
  const float sc = ((const float*)it.g)[r] / nrm;
  float* wr = wbuf + it.w_off + (size_t)r * C;
  for (int i = threadIdx.x; i < C; i += 256) wr[i] = vr[i] * sc;
  if (threadIdx.x == 0) norm[blockIdx.x] = nrm;
}

__global__ void __launch_bounds__(256)
weight_norm_many_bwd_kernel(const WnItem* __restrict__ items, int n, const float* __restrict__ norm) {
  __shared__ float red[4];
  const WnItem it = items[wn_find(items, n, blockIdx.x)];
  const int r = blockIdx.x - (int)it.row0, C = (int)it.C;
  const float* vr = (const float*)it.v + (size_t)r * C;
  const float* dwr = (const float*)it.dw + (size_t)r * C;
  float s = 0.f;
  for (int i = threadIdx.x; i < C; i += 256) s += dwr[i] * vr[i];
  s = block_sum256(s, red);
  const float nrm = norm[blockIdx.x], gg = ((const float*)it.g)[r];
  const float a = gg / nrm, bcoef = gg * s / (nrm * nrm * nrm);
  float* dvr = (float*)it.dv + (size_t)r * C;
  float* dgp = (float*)it.dg + r;
  if (it.acc) {
    for (int i = threadIdx.x; i < C; i += 256) dvr[i] += a * dwr[i] - bcoef * vr[i];
    if (threadIdx.x == 0) *dgp += s / nrm;
  } else {
    for (int i = threadIdx.x; i < C; i += 256) dvr[i] = a * dwr[i] - bcoef * vr[i];
    if (threadIdx.x == 0) *dgp = s / nrm;
  }
}

// ---- small streaming ops --------------------------------------------------------------------
__global__ void avg3_kernel(const float* __restrict__ a, const float* __restrict__ b,
                            const float* __restrict__ c, float* __restrict__ y, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = (a[i] + b[i] + c[i]) / 3.f;
}

__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, float alpha, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = alpha * x[i];
}

// out = dy * act'(y): the activation-derivative mask of a fused conv+activation, applied once so the data
// gradient, weight gradient and bias gradient kernels all stream the same pre-masked tensor
__global__ void act_grad_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                                int tf, float slope, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vcv_tf(dy[i], tf, y, i, slope);
}

// the same with 16-byte accesses and an optional second gradient added first: out = (dy + add) * act'(y).  `add` is the
// gradient a recorded feature map received from the feature-matching loss (ops._TapFn): summed here instead of in a pass of
// its own over the pass-through gradient
template <bool ADD>
__global__ void act_grad4_kernel(const float4* __restrict__ dy, const float4* __restrict__ add, const float4* __restrict__ y,
                                 float4* __restrict__ out, int tf, float slope, size_t n4) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 d = dy[i];
  const float4 a = y[i];
  if (ADD) {
    const float4 e = add[i];
    d.x += e.x; d.y += e.y; d.z += e.z; d.w += e.w;
  }
  out[i] = make_float4(vcv_tf_val(d.x, tf, a.x, slope), vcv_tf_val(d.y, tf, a.y, slope), vcv_tf_val(d.z, tf, a.z, slope),
                       vcv_tf_val(d.w, tf, a.w, slope));
}

__global__ void act_grad_add_kernel(const float* __restrict__ dy, const float* __restrict__ add, const float* __restrict__ y,
                                    float* __restrict__ out, int tf, float slope, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = vcv_tf_val(dy[i] + add[i], tf, y[i], slope);
}

// act_grad over [B, C, T] rows that also collects db[c] += sum_{b, t} out[b, c, t]: work units are 1024-float pieces of
// the contiguous (b, c) rows (as bias_grad_kernel in conv_wgrad.hip); one atomic per (channel, segment)
__global__ void __launch_bounds__(256)
act_grad_bias_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out,
                     float* __restrict__ db, int B, int C, int T, int tf, float slope, int nseg) {
  const int c = blockIdx.x, seg = blockIdx.y;
  const int nchunk = (T + 1023) >> 10;
  const int units = B * nchunk;
  const int per = (units + nseg - 1) / nseg;
  const int lo = seg * per;
  const int hi = lo + per < units ? lo + per : units;
  float s = 0.f;
  // four units per round, all their loads issued before the first store: one unit per iteration was a chain of
  // load -> store latencies (32 units per block on the 1024-channel period layers)
  for (int unit0 = lo; unit0 < hi; unit0 += 4) {
    float dv[4][4], yv[4][4];
    size_t idx[4][4];
    bool in[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int unit = unit0 + q;
      const int b = unit / nchunk, ch = unit - b * nchunk;
      const size_t row = ((size_t)b * C + c) * (size_t)T;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = (ch << 10) + k * 256 + threadIdx.x;
        in[q][k] = unit < hi && t < T;
        idx[q][k] = row + t;
        dv[q][k] = in[q][k] ? dy[row + t] : 0.f;
        yv[q][k] = (in[q][k] && tf != VCV_TF_NONE) ? y[row + t] : 0.f;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (in[q][k]) {
          const float v = vcv_tf_val(dv[q][k], tf, yv[q][k], slope);
          out[idx[q][k]] = v;
          s += v;
        }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(db + c, red[0] + red[1] + red[2] + red[3]);
}

// the same over rows whose length is a multiple of 4, 16 bytes per access: a unit is 1024 floats = one float4 per thread,
// four units (four dy + four y loads) in flight per thread
__global__ void __launch_bounds__(256)
act_grad_bias4_kernel(const float4* __restrict__ dy, const float4* __restrict__ y, float4* __restrict__ out,
                      float* __restrict__ db, int B, int C, int T4, int tf, float slope, int nseg) {
  const int c = blockIdx.x, seg = blockIdx.y;
  const int nchunk = (T4 + 255) >> 8;
  const int units = B * nchunk;
  const int per = (units + nseg - 1) / nseg;
  const int lo = seg * per;
  const int hi = lo + per < units ? lo + per : units;
  float s = 0.f;
  for (int unit0 = lo; unit0 < hi; unit0 += 4) {
    float4 dv[4], yv[4];
    size_t idx[4];
    bool in[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int unit = unit0 + q;
      const int b = unit / nchunk, ch = unit - b * nchunk;
      const int t = (ch << 8) + threadIdx.x;
      in[q] = unit < hi && t < T4;
      idx[q] = ((size_t)b * C + c) * (size_t)T4 + t;
      dv[q] = in[q] ? dy[idx[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
      yv[q] = (in[q] && tf != VCV_TF_NONE) ? y[idx[q]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (in[q]) {
        const float4 v = make_float4(vcv_tf_val(dv[q].x, tf, yv[q].x, slope), vcv_tf_val(dv[q].y, tf, yv[q].y, slope),
                                     vcv_tf_val(dv[q].z, tf, yv[q].z, slope), vcv_tf_val(dv[q].w, tf, yv[q].w, slope));
        out[idx[q]] = v;
        s += (v.x + v.y) + (v.z + v.w);
      }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(db + c, red[0] + red[1] + red[2] + red[3]);
}

// y[b, c, t] = x[b, c, t] * mask[b, t]
__global__ void mask_mul_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                float* __restrict__ y, int C, int T, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t bc = i / T;
  const int t = (int)(i - bc * T);
  const size_t b = bc / C;
  y[i] = x[i] * mask[b * T + t];
}

// right reflect pad of each row: y[r, t] = x[r, t] (t < T), x[r, 2T-2-t] (t >= T)   (F.pad reflect)
__global__ void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int Tp,
                                       size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t r = i / Tp;
  const int t = (int)(i - r * Tp);
  y[i] = x[r * T + (t < T ? t : 2 * T - 2 - t)];
}

__global__ void reflect_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int T, int Tp,
                                       size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t r = i / T;
  const int t = (int)(i - r * T);
  float v = dy[r * Tp + t];
  const int m = 2 * T - 2 - t;  // padded position that mirrors t
  if (m >= T && m < Tp) v += dy[r * Tp + m];
  dx[i] = v;
}

// AvgPool1d(kernel 4, stride 2, padding 2), count_include_pad: To = T/2 + 1
__global__ void avgpool4_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int T, int To,
                                    size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t r = i / To;
  const int o = (int)(i - r * To);
  const float* xr = x + r * T;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int t = 2 * o - 2 + k;
    if (t >= 0 && t < T) s += xr[t];
  }
  y[i] = 0.25f * s;
}

__global__ void avgpool4_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int T, int To,
                                    size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t r = i / T;
  const int t = (int)(i - r * T);
  const float* dr = dy + r * To;
  // outputs o with 2o-2 <= t <= 2o+1  ->  o in {ceil((t-1)/2) .. floor((t+2)/2)}
  float s = 0.f;
  const int o0 = (t + 2) >> 1;
  const int o1 = o0 - 1;
  if (o0 < To) s += dr[o0];
  if (o1 >= 0 && o1 < To) s += dr[o1];
  dx[i] = 0.25f * s;
}

// ---- loss reductions: out[0] += scale * sum(f(a, b)) ------------------------------------------
// mode 0: |a - b|   mode 1: (a - target)^2  (b unused)
__global__ void __launch_bounds__(256)
loss_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, float target, int mode,
                float scale, float* __restrict__ out, size_t n) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (mode == 0) s += fabsf(a[i] - b[i]);
    else { const float d = a[i] - target; s += d * d; }
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out, scale * s);
}

// da = scale * gout[0] * f'(a, b)      (accumulate != 0: da += ...)
__global__ void loss_grad_kernel(const float* __restrict__ a, const float* __restrict__ b, float target,
                                 int mode, float scale, const float* __restrict__ gout,
                                 float* __restrict__ da, int accumulate, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float g = scale * gout[0];
  float v;
  if (mode == 0) { const float d = a[i] - b[i]; v = d > 0.f ? g : (d < 0.f ? -g : 0.f); }
  else v = 2.f * (a[i] - target) * g;
  da[i] = accumulate ? da[i] + v : v;
}

// ---- batched loss terms: one launch for a list of tensors --------------------------------------------------
// items[i] = {a ptr, b ptr (mode 0), n, first block, float bits of scale, offset of da_i (floats) in dabuf}
struct LossItem { long long a, b, n, blk0, scale_bits, da_off; };

__device__ __forceinline__ int loss_find(const LossItem* __restrict__ items, int n, int blk) {
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].blk0 <= blk) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ void __launch_bounds__(256)
loss_many_sum_kernel(const LossItem* __restrict__ items, int nitems, int total_blocks, float target, int mode,
                     float* __restrict__ out) {
  __shared__ float red[4];
  const int ti = loss_find(items, nitems, blockIdx.x);
  const LossItem it = items[ti];
  const int nb = (int)((ti + 1 < nitems ? items[ti + 1].blk0 : total_blocks) - it.blk0);
  const float* a = (const float*)it.a;
  const float* b = (const float*)it.b;
  const size_t n = (size_t)it.n;
  float s = 0.f;
  for (size_t i = (size_t)(blockIdx.x - it.blk0) * 256 + threadIdx.x; i < n; i += (size_t)nb * 256) {
    if (mode == 0) s += fabsf(a[i] - b[i]);
    else { const float d = a[i] - target; s += d * d; }
  }
  s = block_sum256(s, red);
  if (threadIdx.x == 0) unsafeAtomicAdd(out + ti, __int_as_float((int)it.scale_bits) * s);
}

__global__ void __launch_bounds__(256)
loss_many_grad_kernel(const LossItem* __restrict__ items, int nitems, int total_blocks, float target, int mode,
                      const float* __restrict__ gout, float* __restrict__ dabuf) {
  const int ti = loss_find(items, nitems, blockIdx.x);
  const LossItem it = items[ti];
  const int nb = (int)((ti + 1 < nitems ? items[ti + 1].blk0 : total_blocks) - it.blk0);
  const float* a = (const float*)it.a;
  const float* b = (const float*)it.b;
  float* da = dabuf + it.da_off;
  const size_t n = (size_t)it.n;
  const float g = __int_as_float((int)it.scale_bits) * gout[ti];
  for (size_t i = (size_t)(blockIdx.x - it.blk0) * 256 + threadIdx.x; i < n; i += (size_t)nb * 256) {
    float v;
    if (mode == 0) { const float d = a[i] - b[i]; v = d > 0.f ? g : (d < 0.f ? -g : 0.f); }
    else v = 2.f * (a[i] - target) * g;
    da[i] = v;
  }
}

// ---- AdamW over a flat parameter buffer (torch.optim.AdamW semantics) -------------------------
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps,
                             float wd, float bc1, float bc2_sqrt) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i];
  float pi = p[i] * (1.f - lr * wd);
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
}

// The same step with the two per-step scalars read from device memory: hyper = {lr (fp32 bits), step delta (int32)}.  A
// launch recorded into a HIP graph bakes its arguments, so the learning rate and the step count (bias corrections) of a
// replayed optimizer step come from this 8-byte record, which the host refreshes before each replay (light/graphed.py).
__global__ void adamw_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, size_t n, float b1, float b2, float eps, float wd,
                                 const int* __restrict__ hyper, int step_base) {
  __shared__ float s_h[3];
  if (threadIdx.x == 0) {
    const int step = step_base + hyper[1];
    const double bc1 = 1.0 - pow((double)b1, (double)step);
    const double bc2 = 1.0 - pow((double)b2, (double)step);
    s_h[0] = __int_as_float(hyper[0]);
    s_h[1] = (float)bc1;
    s_h[2] = (float)sqrt(bc2);
  }
  __syncthreads();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float lr = s_h[0], bc1 = s_h[1], bc2_sqrt = s_h[2];
  const float gi = g[i];
  float pi = p[i] * (1.f - lr * wd);
  const float mi = b1 * m[i] + (1.f - b1) * gi;
  const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  const float denom = sqrtf(vi) / bc2_sqrt + eps;
  pi -= (lr / bc1) * (mi / denom);
  p[i] = pi; m[i] = mi; v[i] = vi;
}

__global__ void set_words_kernel(int* __restrict__ dst, int n, int w0, int w1, int w2, int w3) {
  const int w[4] = {w0, w1, w2, w3};
  if ((int)threadIdx.x < n) dst[threadIdx.x] = w[threadIdx.x];
}

// wt[c, m, K-1-k] = w[m, c, k]: the stride-1 data gradient of a conv is a forward conv with these weights
__global__ void weight_flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int M, int C,
                                             int K, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int k = (int)(i % K);
  const size_t cm = i / K;
  const int m = (int)(cm % M);
  const int c = (int)(cm / M);
  wt[i] = w[((size_t)m * C + c) * K + (K - 1 - k)];
}

inline dim3 grid1d(size_t n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int vcv_weight_norm_fwd(const float* v, const float* g, float* w, float* norm, int R, int C,
                                   void* stream) {
  if (!v || !g || !w || !norm || R <= 0 || C <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(R), dim3(256), 0, ST, v, g, w, norm, C);
  return vcv_check_launch();
}

extern "C" int vcv_weight_norm_bwd(const float* dw, const float* v, const float* g, const float* norm,
                                   float* dv, float* dg, int R, int C, void* stream) {
  if (!dw || !v || !g || !norm || !dv || !dg || R <= 0 || C <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(R), dim3(256), 0, ST, dw, v, g, norm, dv, dg, C);
  return vcv_check_launch();
}

extern "C" int vcv_weight_norm_many_fwd(const void* items_dev, int n_items, int total_rows, float* wbuf, float* norm,
                                        void* stream) {
  if (!items_dev || !wbuf || !norm || n_items <= 0 || total_rows <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(weight_norm_many_fwd_kernel, dim3(total_rows), dim3(256), 0, ST, (const WnItem*)items_dev, n_items,
                     wbuf, norm);
  return vcv_check_launch();
}

extern "C" int vcv_weight_norm_many_bwd(const void* items_dev, int n_items, int total_rows, const float* norm,
                                        void* stream) {
  if (!items_dev || !norm || n_items <= 0 || total_rows <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(weight_norm_many_bwd_kernel, dim3(total_rows), dim3(256), 0, ST, (const WnItem*)items_dev, n_items,
                     norm);
  return vcv_check_launch();
}

extern "C" int vcv_weight_flip_transpose(const float* w, float* wt, int M, int C, int K, void* stream) {
  if (!w || !wt || M <= 0 || C <= 0 || K <= 0) return VCV_EINVAL;
  const size_t n = (size_t)M * C * K;
  hipLaunchKernelGGL(weight_flip_transpose_kernel, grid1d(n), dim3(256), 0, ST, w, wt, M, C, K, n);
  return vcv_check_launch();
}

extern "C" int vcv_act_grad_add(const float* dy, const float* add, const float* y, float* out, int tf, float slope, int64_t n,
                                void* stream) {
  if (!dy || !y || !out || n <= 0 || tf < VCV_TF_DLEAKY) return VCV_EINVAL;
  const bool vec = (n & 3) == 0 && (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out | (uintptr_t)add) & 15) == 0;
  const bool scalar_only = !vcv_tuning().act_grad_vec;  // (A/B switch)
  if (vec && !scalar_only) {
    const size_t n4 = (size_t)n / 4;
    if (add)
      hipLaunchKernelGGL(act_grad4_kernel<true>, grid1d(n4), dim3(256), 0, ST, (const float4*)dy, (const float4*)add, (const float4*)y,
                         (float4*)out, tf, slope, n4);
    else
      hipLaunchKernelGGL(act_grad4_kernel<false>, grid1d(n4), dim3(256), 0, ST, (const float4*)dy, (const float4*)nullptr,
                         (const float4*)y, (float4*)out, tf, slope, n4);
  } else if (add) {
    hipLaunchKernelGGL(act_grad_add_kernel, grid1d(n), dim3(256), 0, ST, dy, add, y, out, tf, slope, (size_t)n);
  } else {
    hipLaunchKernelGGL(act_grad_kernel, grid1d(n), dim3(256), 0, ST, dy, y, out, tf, slope, (size_t)n);
  }
  return vcv_check_launch();
}

extern "C" int vcv_act_grad(const float* dy, const float* y, float* out, int tf, float slope, int64_t n, void* stream) {
  return vcv_act_grad_add(dy, nullptr, y, out, tf, slope, n, stream);
}

extern "C" int vcv_act_grad_bias(const float* dy, const float* y, float* out, float* dbias, int B, int C, int T, int tf,
                                 float slope, void* stream) {
  if (!dy || !y || !out || !dbias || B <= 0 || C <= 0 || T <= 0 || tf < VCV_TF_DLEAKY) return VCV_EINVAL;
  const long long units = (long long)B * ((T + 1023) / 1024);
  long long nseg = (2048 + C - 1) / C;
  if (nseg > units) nseg = units;
  if (nseg < 1 || vcv_get_deterministic()) nseg = 1;
  const bool scalar_only = !vcv_tuning().act_grad_vec;  // (A/B switch)
  if (!scalar_only && (T & 3) == 0 && (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)out) & 15) == 0)
    hipLaunchKernelGGL(act_grad_bias4_kernel, dim3(C, (unsigned)nseg), dim3(256), 0, ST, (const float4*)dy, (const float4*)y,
                       (float4*)out, dbias, B, C, T / 4, tf, slope, (int)nseg);
  else
    hipLaunchKernelGGL(act_grad_bias_kernel, dim3(C, (unsigned)nseg), dim3(256), 0, ST, dy, y, out, dbias, B, C, T, tf, slope,
                       (int)nseg);
  return vcv_check_launch();
}

extern "C" int vcv_avg3(const float* a, const float* b, const float* c, float* y, int64_t n, void* stream) {
  if (n <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(avg3_kernel, grid1d(n), dim3(256), 0, ST, a, b, c, y, (size_t)n);
  return vcv_check_launch();
}

extern "C" int vcv_scale(const float* x, float* y, float alpha, int64_t n, void* stream) {
  if (n <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(scale_kernel, grid1d(n), dim3(256), 0, ST, x, y, alpha, (size_t)n);
  return vcv_check_launch();
}

extern "C" int vcv_mask_mul(const float* x, const float* mask, float* y, int B, int C, int T, void* stream) {
  const size_t n = (size_t)B * C * T;
  if (n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(mask_mul_kernel, grid1d(n), dim3(256), 0, ST, x, mask, y, C, T, n);
  return vcv_check_launch();
}

extern "C" int vcv_reflect_pad_fwd(const float* x, float* y, int R, int T, int Tp, void* stream) {
  if (R <= 0 || T <= 1 || Tp < T || Tp - T > T - 1) return VCV_EINVAL;
  const size_t n = (size_t)R * Tp;
  hipLaunchKernelGGL(reflect_pad_fwd_kernel, grid1d(n), dim3(256), 0, ST, x, y, T, Tp, n);
  return vcv_check_launch();
}

extern "C" int vcv_reflect_pad_bwd(const float* dy, float* dx, int R, int T, int Tp, void* stream) {
  if (R <= 0 || T <= 1 || Tp < T || Tp - T > T - 1) return VCV_EINVAL;
  const size_t n = (size_t)R * T;
  hipLaunchKernelGGL(reflect_pad_bwd_kernel, grid1d(n), dim3(256), 0, ST, dy, dx, T, Tp, n);
  return vcv_check_launch();
}

extern "C" int vcv_avgpool4_fwd(const float* x, float* y, int R, int T, void* stream) {
  if (R <= 0 || T <= 0) return VCV_EINVAL;
  const int To = T / 2 + 1;
  const size_t n = (size_t)R * To;
  hipLaunchKernelGGL(avgpool4_fwd_kernel, grid1d(n), dim3(256), 0, ST, x, y, T, To, n);
  return vcv_check_launch();
}

extern "C" int vcv_avgpool4_bwd(const float* dy, float* dx, int R, int T, void* stream) {
  if (R <= 0 || T <= 0) return VCV_EINVAL;
  const int To = T / 2 + 1;
  const size_t n = (size_t)R * T;
  hipLaunchKernelGGL(avgpool4_bwd_kernel, grid1d(n), dim3(256), 0, ST, dy, dx, T, To, n);
  return vcv_check_launch();
}

extern "C" int vcv_loss_sum(const float* a, const float* b, float target, int mode, float scale,
                            float* out, int64_t n, void* stream) {
  if (!a || !out || n <= 0 || (mode == 0 && !b)) return VCV_EINVAL;
  size_t nb = ((size_t)n + 2047) / 2048;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(loss_sum_kernel, dim3((unsigned)nb), dim3(256), 0, ST, a, b, target, mode, scale,
                     out, (size_t)n);
  return vcv_check_launch();
}

extern "C" int vcv_loss_grad(const float* a, const float* b, float target, int mode, float scale,
                             const float* gout, float* da, int accumulate, int64_t n, void* stream) {
  if (!a || !gout || !da || n <= 0 || (mode == 0 && !b)) return VCV_EINVAL;
  hipLaunchKernelGGL(loss_grad_kernel, grid1d(n), dim3(256), 0, ST, a, b, target, mode, scale, gout, da,
                     accumulate, (size_t)n);
  return vcv_check_launch();
}

extern "C" int vcv_loss_many_sum(const void* items_dev, int n_items, int total_blocks, float target, int mode, float* out,
                                 void* stream) {
  if (!items_dev || !out || n_items <= 0 || total_blocks <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(loss_many_sum_kernel, dim3(total_blocks), dim3(256), 0, ST, (const LossItem*)items_dev, n_items,
                     total_blocks, target, mode, out);
  return vcv_check_launch();
}

extern "C" int vcv_loss_many_grad(const void* items_dev, int n_items, int total_blocks, float target, int mode,
                                  const float* gout, float* dabuf, void* stream) {
  if (!items_dev || !gout || !dabuf || n_items <= 0 || total_blocks <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(loss_many_grad_kernel, dim3(total_blocks), dim3(256), 0, ST, (const LossItem*)items_dev, n_items,
                     total_blocks, target, mode, gout, dabuf);
  return vcv_check_launch();
}

// dst[0..n) = the first n of (w0, w1, w2, w3), n <= 4: a few host scalars into device memory as kernel ARGUMENTS (copied
// at launch time: no host buffer that a later call could overwrite while the copy is still queued).
extern "C" int vcv_set_words(void* dst, int n, int w0, int w1, int w2, int w3, void* stream) {
  if (!dst || n <= 0 || n > 4) return VCV_EINVAL;
  hipLaunchKernelGGL(set_words_kernel, dim3(1), dim3(64), 0, ST, (int*)dst, n, w0, w1, w2, w3);
  return vcv_check_launch();
}

extern "C" int vcv_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, float b1, float b2, float eps,
                             float wd, const void* hyper_dev, int step_base, void* stream) {
  if (!p || !g || !m || !v || !hyper_dev || n <= 0 || step_base <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(adamw_dev_kernel, grid1d(n), dim3(256), 0, ST, p, g, m, v, (size_t)n, b1, b2, eps, wd,
                     (const int*)hyper_dev, step_base);
  return vcv_check_launch();
}

extern "C" int vcv_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1,
                         float b2, float eps, float wd, int step, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return VCV_EINVAL;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, grid1d(n), dim3(256), 0, ST, p, g, m, v, (size_t)n, lr, b1, b2, eps, wd,
                     (float)bc1, (float)sqrt(bc2));
  return vcv_check_launch();
}

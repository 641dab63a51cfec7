// conv_thin.hip -- convolutions with ONE output channel or ONE input channel per group.
//
// The discriminators end in Conv(1024 -> 1, k3) and start with Conv(1 -> 16/32) (reference:
// discriminator.py:18,25,53,61), the generator ends in Conv(32 -> 1, k7) + tanh.
// These are matrix-vector shaped and HBM-bound: padding them into 32-row MFMA tiles wastes the
// matrix pipe and, worse, serialises a 3072-deep reduction inside a handful of workgroups.  Here:
//   conv_m1_fwd   y[b,t,p] = act(bias + sum_{c,k} w[c,k] * tf(x[b,c,t*s+k*d-pad,p])): lanes along the
//                 contiguous positions (coalesced row reads), the channel range split over the grid
//                 and combined with fp32 atomics when it is deep;
//   thin_wgrad    dw[m,c,k] for min(M,C) == 1: one workgroup per (m,c) weight row, lanes along the
//                 positions, K accumulators per lane, wavefront-shuffle reduction.
#include "common.h"

namespace {

__device__ __forceinline__ float wsum(float s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  return s;
}

constexpr int KMAX = 16;

// grid: (position tiles of 256, B, channel splits)
__global__ void __launch_bounds__(256)
conv_m1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ y, int C, int Tin, int Tout, int P, int K, int s, int d, int pad,
                   int in_leaky, int out_act, float slope, int cper, int atomic) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  const int c_lo = blockIdx.z * cper;
  int c_hi = c_lo + cper;
  if (c_hi > C) c_hi = C;
  const int U = Tout * P;
  if (u >= U) return;
  const int t = u / P, pc = u - t * P;
  const long long TinP = (long long)Tin * P;
  const float* xb = x + (size_t)b * C * TinP;
  // the tap offsets inside a channel row do not depend on the channel: hoisted (0 with weight 0 outside the row)
  int offk[KMAX];
  float okk[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int r = t * s + k * d - pad;
    const bool ok = k < K && r >= 0 && r < Tin;
    offk[k] = ok ? r * P + pc : 0;
    okk[k] = ok ? 1.f : 0.f;
  }
  float acc = 0.f;
#pragma unroll 4
  for (int c = c_lo; c < c_hi; ++c) {
    const float* xr = xb + (size_t)c * TinP;
    const float* wr = w + (size_t)c * K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        float v = xr[offk[k]];
        if (in_leaky) v = vcv_leaky(v, slope);
        acc += wr[k] * okk[k] * v;
      }
    }
  }
  const size_t oi = (size_t)b * U + u;
  if (atomic) {
    unsafeAtomicAdd(y + oi, acc);
  } else {
    if (bias) acc += bias[0];
    y[oi] = vcv_act(acc, out_act, slope);
  }
}

// stride 1: the input block of a (batch element, position tile, channel chunk) is staged in LDS with independent
// coalesced loads (eight rows in flight per lane), then every lane reduces its position over the chunk out of
// LDS.  The register version above issues C*K dependent-latency loads per lane and is latency-bound on the
// discriminators' Conv(1024 -> 1, k3) heads (rows of 30..250 positions) and on the generator's Conv(32 -> 1, k7).
// Short rows: R = 256 / TU channel sub-rows share a workgroup and meet in LDS.
// grid: (position tiles of TU, B, channel splits); dynamic LDS: cch * (W + K) + 256 floats
__global__ void __launch_bounds__(256)
conv_m1_lds_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ y, int C, int Uin, int U, int K, int tapstep, int shift, int in_leaky,
                   int out_act, float slope, int cper, int cch, int TU, int R, int W, int atomic, int flat) {
  extern __shared__ float sm[];
  float* xs = sm;             // [cch][W]
  float* wsm = sm + cch * W;  // [cch][K]
  float* red = wsm + cch * K; // [256]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int u0 = blockIdx.x * TU;
  const int c_lo = blockIdx.z * cper;
  const int c_hi = c_lo + cper < C ? c_lo + cper : C;
  const int cl = tid / TU, ul = tid - cl * TU;
  const bool live = cl < R && u0 + ul < U;
  const int in_lo = u0 - shift;
  const int j0 = tid, j1 = tid + 256;  // W <= 512 (checked by the launcher)
  const int ui0 = in_lo + j0, ui1 = in_lo + j1;
  const bool ok0 = j0 < W && ui0 >= 0 && ui0 < Uin, ok1 = j1 < W && ui1 >= 0 && ui1 < Uin;
  float acc = 0.f;
  for (int c0 = c_lo; c0 < c_hi; c0 += cch) {
    const int nc = cch < c_hi - c0 ? cch : c_hi - c0;
    const float* xb = x + ((size_t)b * C + c0) * Uin;
    if (flat) {
      // the tile holds whole rows: the chunk's rows are one contiguous block -- a flat copy, eight loads in flight
      const int tot = nc * Uin, hw = W - Uin;
      for (int i = tid; i < nc * hw; i += 256) {
        const int c = i / hw, h = i - c * hw;
        xs[c * W + (h < shift ? h : Uin + h)] = 0.f;
      }
      for (int i0 = tid; i0 < tot; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = i0 + i * 256 < tot ? xb[i0 + i * 256] : 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int idx = i0 + i * 256;
          if (idx < tot) {
            const int c = idx / Uin, ui = idx - c * Uin;
            xs[c * W + shift + ui] = in_leaky ? vcv_leaky(v[i], slope) : v[i];
          }
        }
      }
    } else
    for (int r0 = 0; r0 < nc; r0 += 8) {
      float v0[8], v1[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool rok = r0 + i < nc;
        const float* xr = xb + (size_t)(r0 + i) * Uin;
        v0[i] = rok && ok0 ? xr[ui0] : 0.f;
        v1[i] = rok && ok1 ? xr[ui1] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (r0 + i < nc) {
          if (j0 < W) xs[(r0 + i) * W + j0] = in_leaky ? vcv_leaky(v0[i], slope) : v0[i];
          if (j1 < W) xs[(r0 + i) * W + j1] = in_leaky ? vcv_leaky(v1[i], slope) : v1[i];
        }
      }
    }
    for (int i = tid; i < nc * K; i += 256) wsm[i] = w[(size_t)c0 * K + i];
    __syncthreads();
    if (live) {
      for (int c = cl; c < nc; c += R) {
        const float* xr = xs + c * W + ul;
        const float* wr = wsm + c * K;
        for (int k = 0; k < K; ++k) acc += wr[k] * xr[k * tapstep];
      }
    }
    __syncthreads();
  }
  if (R > 1) {
    red[tid] = live ? acc : 0.f;
    __syncthreads();
    if (cl == 0 && live)
      for (int r = 1; r < R; ++r) acc += red[r * TU + ul];
  }
  if (cl == 0 && live) {
    const size_t oi = (size_t)b * U + u0 + ul;
    if (atomic) {
      unsafeAtomicAdd(y + oi, acc);
    } else {
      if (bias) acc += bias[0];
      y[oi] = vcv_act(acc, out_act, slope);
    }
  }
}

__global__ void fill_bias_kernel(float* __restrict__ y, const float* __restrict__ bias, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = bias ? bias[0] : 0.f;
}

// one workgroup per weight row (m, c); a: [B, M, Ta, P] (un-shifted), bsh: [B, C, Tb, P] (shifted)
// PLAIN: no operand transforms (the usual case: dy arrives pre-masked) -- no per-element transform switch
// KT > 0: the tap count at compile time -- the tap loop is then straight-line code, its loads are issued together and
// out-of-row taps are clamped reads with a zero weight instead of branches
template <bool PLAIN, int KT>
__global__ void __launch_bounds__(256)
thin_wgrad_kernel(const float* __restrict__ a, const float* __restrict__ bsh, const float* __restrict__ aaux,
                  const float* __restrict__ baux, float* __restrict__ dw, int B, int M, int C, int Ta, int Tb,
                  int P, int K, int s, int d, int off, int a_tf, int b_tf, float slope, float alpha, int bper, int uper) {
  constexpr int KK = KT > 0 ? KT : KMAX;
  __shared__ float red[4][KMAX];
  const int m = blockIdx.x / C, c = blockIdx.x % C;
  const int U = Ta * P;  // < 2^31 (checked by the launcher)
  float acc[KK];
#pragma unroll
  for (int k = 0; k < KK; ++k) acc[k] = 0.f;
  const int b_lo = blockIdx.y * bper;
  int b_hi = b_lo + bper;
  if (b_hi > B) b_hi = B;
  const int u_lo0 = blockIdx.z * uper, ulen = (u_lo0 + uper < U ? u_lo0 + uper : U) - u_lo0;
  if (ulen < 512) {
    // short rows (the discriminator heads: 2..250 positions): lanes along the flattened (batch element, position)
    // range, so that a wavefront is full and the loop is a few independent iterations instead of one per element
    const int tot = (b_hi - b_lo) * ulen;
    const int kstep = d * P;
#pragma unroll 4
    for (int j = threadIdx.x; j < tot; j += 256) {
      const int bl = j / ulen, u = u_lo0 + j - bl * ulen, b = b_lo + bl;
      const size_t abase = ((size_t)b * M + m) * (size_t)U;
      const size_t bbase = ((size_t)b * C + c) * (size_t)Tb * P;
      const float* brow = bsh + bbase;
      float av = a[abase + u];
      if (!PLAIN) av = vcv_tf(av, a_tf, aaux, abase + u, slope);
      const int q = u / P, pc = u - q * P;
      const int r0 = q * s + off;
      const int i0 = r0 * P + pc;
      if (PLAIN && KT > 0) {
        float bv[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) {
          const int r = r0 + k * d;
          const bool ok = r >= 0 && r < Tb;
          bv[k] = brow[ok ? i0 + k * kstep : 0];
          bv[k] = ok ? bv[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) acc[k] += av * bv[k];
      } else {
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        if (k < K) {
          const int r = r0 + k * d;
          if (r >= 0 && r < Tb) {
            const int bi = i0 + k * kstep;
            const float bv = PLAIN ? brow[bi] : vcv_tf(brow[bi], b_tf, baux, bbase + bi, slope);
            acc[k] += av * bv;
          }
        }
      }
      }
    }
  } else
  for (int b = b_lo; b < b_hi; ++b) {
    const size_t abase = ((size_t)b * M + m) * (size_t)U;
    const size_t bbase = ((size_t)b * C + c) * (size_t)Tb * P;
    const int u_lo = blockIdx.z * uper, u_hi = u_lo + uper < U ? u_lo + uper : U;
    const float* brow = bsh + bbase;
    const int kstep = d * P;
#pragma unroll 4
    for (int u = u_lo + threadIdx.x; u < u_hi; u += 256) {
      float av = a[abase + u];
      if (!PLAIN) av = vcv_tf(av, a_tf, aaux, abase + u, slope);
      const int q = u / P, pc = u - q * P;
      const int r0 = q * s + off;
      const int i0 = r0 * P + pc;  // element offset of tap 0 inside the (b, c) row (fits 32 bits: checked by the launcher)
      if (PLAIN && KT > 0) {
        float bv[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) {
          const int r = r0 + k * d;
          const bool ok = r >= 0 && r < Tb;
          bv[k] = brow[ok ? i0 + k * kstep : 0];
          bv[k] = ok ? bv[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) acc[k] += av * bv[k];
      } else {
#pragma unroll
      for (int k = 0; k < KK; ++k) {
        if (k < K) {
          const int r = r0 + k * d;
          if (r >= 0 && r < Tb) {
            const int bi = i0 + k * kstep;
            const float bv = PLAIN ? brow[bi] : vcv_tf(brow[bi], b_tf, baux, bbase + bi, slope);
            acc[k] += av * bv;
          }
        }
      }
      }
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KK; ++k) {
    if (k >= K) break;  // uniform
    const float v = wsum(acc[k]);
    if (lane == 0) red[wv][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    unsafeAtomicAdd(dw + ((size_t)m * C + c) * K + k, alpha * (red[0][k] + red[1][k] + red[2][k] + red[3][k]));
  }
}

// ---- weight gradient of a one-input-channel layer with M*K <= 256 weights: lane = one weight (m, k) ---------
// dw[m,k] = sum_{b,q,p} dy[b,m,q,p] * x[b,0,q*s+off+k*d,p].  The workgroup walks a (batch element, row chunk): the
// x window of the chunk and [M][tile] slices of dy are staged in LDS with coalesced loads, and every position then
// costs a lane two LDS reads and one FMA -- no per-element tap loads from memory (K+1 loads per element made the
// row-per-workgroup kernel above texture-address-bound at K = 15) and no cross-lane reduction.
// grid: (row chunks of qper, B); dynamic LDS: xw + M*TUP floats
__global__ void __launch_bounds__(256)
c1_wgrad_pairs_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int M, int Ta,
                      int Tb, int P, int K, int s, int d, int off, float alpha, int qper, int tq, int TUP, int xw) {
  extern __shared__ float sm[];
  float* xs = sm;        // [xw]: rows q_lo*s+off .. of x[b], zero outside the row
  float* dyt = sm + xw;  // [M][TUP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int q_lo = blockIdx.x * qper;
  const int q_hi = q_lo + qper < Ta ? q_lo + qper : Ta;
  const int m = tid / K, k = tid - m * K;
  const bool own = tid < M * K;
  {
    const long long lo = (long long)(q_lo * s + off) * P, n = (long long)Tb * P;
    const float* xb = x + (size_t)b * n;
    for (int i = tid; i < xw; i += 256) {
      const long long g = lo + i;
      xs[i] = g >= 0 && g < n ? xb[g] : 0.f;
    }
  }
  float acc = 0.f;
  const size_t U = (size_t)Ta * P;
  const float* ar = dyt + (own ? m : 0) * TUP;
  for (int q0 = q_lo; q0 < q_hi; q0 += tq) {
    const int nq = tq < q_hi - q0 ? tq : q_hi - q0;
    const int tl = nq * P;
    __syncthreads();  // the previous tile is consumed (and, first time round, xs is written)
    for (int mm = wave; mm < M; mm += 4) {
      const float* src = dy + ((size_t)b * M + mm) * U + (size_t)q0 * P;
      for (int j = lane; j < tl; j += 64) dyt[mm * TUP + j] = src[j];
    }
    __syncthreads();
    if (own) {
      const float* xr = xs + ((q0 - q_lo) * s + k * d) * P;
      if (s == 1) {
#pragma unroll 8
        for (int j = 0; j < tl; ++j) acc += ar[j] * xr[j];
      } else {
        for (int ql = 0; ql < nq; ++ql) {
          const float* a2 = ar + ql * P;
          const float* x2 = xr + ql * s * P;
#pragma unroll 4
          for (int pc = 0; pc < P; ++pc) acc += a2[pc] * x2[pc];
        }
      }
    }
  }
  if (own) unsafeAtomicAdd(dw + tid, alpha * acc);
}

// ---- one INPUT channel (first layers of the discriminators: 1 -> 16 k15, 1 -> 32 k5 stride 3) -------------
// forward: each lane holds the K input taps of its position in registers and produces all M output channels
// (M*K FMAs per position, coalesced row writes) -- the layer is an HBM write stream, not a GEMM.
constexpr int C1_MMAX = 64;

__global__ void __launch_bounds__(256)
conv_c1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ y, int M, int Tin, int Tout, int P, int K, int s, int d, int pad, int out_act,
                   float slope, int mch, const float* __restrict__ oaux, int flip) {
  __shared__ float ws[C1_MMAX * KMAX];
  __shared__ float bs[C1_MMAX];
  // blockIdx.z: chunk of mch <= C1_MMAX output channels (wide one-input-channel layers, e.g. the data gradient of a
  // 1024 -> 1 conv_post, which is a one-input-channel convolution with flipped taps; narrow chunks when the position
  // tiles alone leave the chip short of workgroups -- the layer is a write stream of M rows per position tile)
  const int mfull = M, m0 = blockIdx.z * mch;
  M = M - m0 < mch ? M - m0 : mch;
  w += (size_t)m0 * K;
  if (bias) bias += m0;
  // (flip: taps in reverse order -- the data gradient of a one-OUTPUT-channel conv reads its [C, K] weight rows backwards;
  // done here while staging instead of by a flipped copy made with a launch of its own, 28 of them per training step)
  for (int i = threadIdx.x; i < M * K; i += 256) ws[i] = flip ? w[i - (i % K) + (K - 1 - i % K)] : w[i];
  for (int i = threadIdx.x; i < M; i += 256) bs[i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int U = Tout * P;
  // rows shorter than a workgroup (the heads' data gradients: 30..250 positions, 1024 channels): R = 256 / U lanes share a
  // position and take every R-th channel of the chunk
  const int TU = U < 256 ? U : 256, R = 256 / TU;
  const int mr = threadIdx.x / TU;
  const int u = blockIdx.x * TU + threadIdx.x - mr * TU;
  if (u >= U || mr >= R) return;
  const int b = blockIdx.y;
  const int t = u / P, pc = u - t * P;
  const float* xb = x + (size_t)b * Tin * P;
  float xv[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int r = t * s + k * d - pad;
    xv[k] = (k < K && r >= 0 && r < Tin) ? xb[(size_t)r * P + pc] : 0.f;
  }
  float* yb = y + ((size_t)b * mfull + m0) * U + u;
  // oaux (a data gradient's launch): the leaky-ReLU output this gradient flows back into -- its derivative in the epilogue
  const float* ob = oaux ? oaux + ((size_t)b * mfull + m0) * U + u : nullptr;
  for (int m = mr; m < M; m += R) {
    float acc = bs[m];
    const float* wr = ws + m * K;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (k < K) acc += wr[k] * xv[k];
    acc = vcv_act(acc, out_act, slope);
    if (ob) acc *= vcv_dleaky(ob[(size_t)m * U], slope);
    yb[(size_t)m * U] = acc;
  }
}

// data gradient w.r.t. the single input channel: dx[b,t,p] = sum_{m,k} w[m,k] * dy[b,m,q,p], q*s + k*d - pad == t
__global__ void __launch_bounds__(256)
conv_c1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int M, int Tin,
                     int Tout, int P, int K, int s, int d, int pad) {
  __shared__ float ws[C1_MMAX * KMAX];
  for (int i = threadIdx.x; i < M * K; i += 256) ws[i] = w[i];
  __syncthreads();
  const int UI = Tin * P;
  const int u = blockIdx.x * 256 + threadIdx.x;
  if (u >= UI) return;
  const int b = blockIdx.y;
  const int t = u / P, pc = u - t * P;
  const size_t UO = (size_t)Tout * P;
  const float* dyb = dy + (size_t)b * M * UO;
  float acc = 0.f;
  for (int k = 0; k < K; ++k) {
    const int r = t + pad - k * d;
    if (r < 0 || r % s != 0) continue;
    const int q = r / s;
    if (q >= Tout) continue;
    const float* dp = dyb + (size_t)q * P + pc;
    for (int m = 0; m < M; ++m) acc += ws[m * K + k] * dp[(size_t)m * UO];
  }
  dx[(size_t)b * UI + u] = acc;
}

}  // namespace

// ---- one-frame pointwise layers (speaker conditioning: Conv1d(gin, 2*H*L, 1) on g [B, gin, 1]) ------------
// A [M, C] matrix against B <= 32 vectors: HBM-bound on the weight matrix, far too few columns for a GEMM tile.
constexpr int T1_BMAX = 32;

namespace {
// y[b, m] = bias[m] + sum_c w[m, c] * x[b, c]: one wavefront per output row, lanes along c
__global__ void __launch_bounds__(256)
linear_t1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                     float* __restrict__ y, int B, int C, int M) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  float acc[T1_BMAX];
#pragma unroll
  for (int b = 0; b < T1_BMAX; ++b) acc[b] = 0.f;
  const float* wr = w + (size_t)m * C;
  for (int c = lane; c < C; c += 64) {
    const float wv = wr[c];
#pragma unroll
    for (int b = 0; b < T1_BMAX; ++b)
      if (b < B) acc[b] += wv * x[(size_t)b * C + c];
  }
  const float bv = bias ? bias[m] : 0.f;
#pragma unroll
  for (int b = 0; b < T1_BMAX; ++b) {
    if (b < B) {
      const float s = wsum(acc[b]);
      if (lane == 0) y[(size_t)b * M + m] = s + bv;
    }
  }
}

// dx[b, c] = sum_m w[m, c] * dy[b, m]: threads along c, the rows split over grid.y and combined with atomics
__global__ void __launch_bounds__(256)
linear_t1_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int B, int C,
                       int M, int mper) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int m_lo = blockIdx.y * mper, m_hi = m_lo + mper < M ? m_lo + mper : M;
  if (c >= C) return;
  float acc[T1_BMAX];
#pragma unroll
  for (int b = 0; b < T1_BMAX; ++b) acc[b] = 0.f;
  for (int m = m_lo; m < m_hi; ++m) {
    const float wv = w[(size_t)m * C + c];
#pragma unroll
    for (int b = 0; b < T1_BMAX; ++b)
      if (b < B) acc[b] += wv * dy[(size_t)b * M + m];  // wave-uniform address: one broadcast load
  }
#pragma unroll
  for (int b = 0; b < T1_BMAX; ++b)
    if (b < B) unsafeAtomicAdd(dx + (size_t)b * C + c, acc[b]);
}

// dw[m, c] += sum_b dy[b, m] * x[b, c]
__global__ void __launch_bounds__(256)
linear_t1_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, int B, int C,
                       int M) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)M * C) return;
  const int m = (int)(i / C), c = (int)(i - (size_t)m * C);
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += dy[(size_t)b * M + m] * x[(size_t)b * C + c];
  dw[i] += s;
}
}  // namespace

extern "C" int vcv_linear_t1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int M,
                                 void* stream) {
  if (!x || !w || !y || B <= 0 || B > T1_BMAX || C <= 0 || M <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(linear_t1_fwd_kernel, dim3(vcv_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, B, C, M);
  return vcv_check_launch();
}

extern "C" int vcv_linear_t1_dgrad(const float* dy, const float* w, float* dx, int B, int C, int M, void* stream) {
  if (!dy || !w || !dx || B <= 0 || B > T1_BMAX || C <= 0 || M <= 0) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (vcv_zero_async(dx, sizeof(float) * (size_t)B * C, st) != hipSuccess) return VCV_EHIP;
  const int nct = vcv_cdiv(C, 256);
  int msplit = 512 / nct;
  if (msplit > M / 16) msplit = M / 16;
  if (msplit < 1) msplit = 1;
  const int mper = vcv_cdiv(M, msplit);
  hipLaunchKernelGGL(linear_t1_dgrad_kernel, dim3(nct, vcv_cdiv(M, mper)), dim3(256), 0, st, dy, w, dx, B, C, M, mper);
  return vcv_check_launch();
}

extern "C" int vcv_linear_t1_wgrad(const float* dy, const float* x, float* dw, int B, int C, int M, void* stream) {
  if (!dy || !x || !dw || B <= 0 || C <= 0 || M <= 0) return VCV_EINVAL;
  const size_t n = (size_t)M * C;
  hipLaunchKernelGGL(linear_t1_wgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, x, dw,
                     B, C, M);
  return vcv_check_launch();
}

extern "C" int vcv_conv_c1_fwd_masked(const float* x, const float* w, const float* bias, float* y, const float* oaux, int B,
                                      int M, int Tin, int Tout, int P, int K, int stride, int dil, int pad, int out_act,
                                      float slope, void* stream);
extern "C" int vcv_conv_c1_fwd_flip(const float* x, const float* w, const float* bias, float* y, const float* oaux, int B,
                                    int M, int Tin, int Tout, int P, int K, int stride, int dil, int pad, int out_act,
                                    float slope, int flip_taps, void* stream);
extern "C" int vcv_conv_c1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int M, int Tin,
                               int Tout, int P, int K, int stride, int dil, int pad, int out_act, float slope,
                               void* stream) {
  return vcv_conv_c1_fwd_masked(x, w, bias, y, nullptr, B, M, Tin, Tout, P, K, stride, dil, pad, out_act, slope, stream);
}

extern "C" int vcv_conv_c1_fwd_masked(const float* x, const float* w, const float* bias, float* y, const float* oaux, int B,
                                      int M, int Tin, int Tout, int P, int K, int stride, int dil, int pad, int out_act,
                                      float slope, void* stream) {
  return vcv_conv_c1_fwd_flip(x, w, bias, y, oaux, B, M, Tin, Tout, P, K, stride, dil, pad, out_act, slope, 0, stream);
}

extern "C" int vcv_conv_c1_fwd_flip(const float* x, const float* w, const float* bias, float* y, const float* oaux, int B,
                                    int M, int Tin, int Tout, int P, int K, int stride, int dil, int pad, int out_act,
                                    float slope, int flip_taps, void* stream) {
  if (!x || !w || !y || B <= 0 || M <= 0 || Tin <= 0 || Tout <= 0 || P <= 0 || K <= 0 || K > KMAX || stride <= 0)
    return VCV_EINVAL;
  // channels per workgroup: halve the chunk (64 .. 8) until the grid has ~2,048 workgroups
  const int force = vcv_tuning().c1_chunk;  // (A/B switch)
  int mch = M < C1_MMAX ? M : C1_MMAX;
  const long long tiles = (long long)vcv_cdiv(Tout * P, 256) * B;
  // (measured, kernel-only: the 1 -> 32 first layers 19.8 -> 16.7 us, the pooled 1 -> 16 ones 15.2 -> 11.2; rows shorter
  // than a position tile -- the 1024-row data gradients of the heads -- are level at 32 and slower below)
  const int floor_ch = Tout * P < 256 ? 32 : 8;
  while (mch > floor_ch && tiles * vcv_cdiv(M, mch) < 2048) mch = (mch + 1) / 2;
  if (force > 0) mch = force < C1_MMAX ? force : C1_MMAX;
  hipLaunchKernelGGL(conv_c1_fwd_kernel, dim3(vcv_cdiv(Tout * P, 256), B, vcv_cdiv(M, mch)), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     y, M, Tin, Tout, P, K, stride, dil, pad, out_act, slope, mch, oaux, flip_taps ? 1 : 0);
  return vcv_check_launch();
}

extern "C" int vcv_conv_c1_dgrad(const float* dy, const float* w, float* dx, int B, int M, int Tin, int Tout, int P,
                                 int K, int stride, int dil, int pad, void* stream) {
  if (!dy || !w || !dx || B <= 0 || M <= 0 || M > C1_MMAX || Tin <= 0 || Tout <= 0 || P <= 0 || K <= 0 || K > KMAX ||
      stride <= 0)
    return VCV_EINVAL;
  hipLaunchKernelGGL(conv_c1_dgrad_kernel, dim3(vcv_cdiv(Tin * P, 256), B), dim3(256), 0, (hipStream_t)stream, dy, w, dx,
                     M, Tin, Tout, P, K, stride, dil, pad);
  return vcv_check_launch();
}

extern "C" int vcv_conv_m1_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C,
                               int Tin, int Tout, int P, int K, int stride, int dil, int pad, int in_leaky,
                               int out_act, float slope, void* stream) {
  if (!x || !w || !y || B <= 0 || C <= 0 || Tin <= 0 || Tout <= 0 || P <= 0 || K <= 0 || stride <= 0) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int U = Tout * P;
  const int nt = vcv_cdiv(U, 256);
  // split the channel range until the grid has ~1000 workgroups
  int splits = 1;
  if (out_act == VCV_ACT_NONE) {
    while (splits < 64 && (long long)nt * B * splits < 1024 && C / (splits * 2) >= 8) splits *= 2;
  }
  if (vcv_get_deterministic()) splits = 1;  // (the channel-range splits meet in fp32 atomics)
  const bool no_lds = !vcv_tuning().m1_lds;
  const long long halo = (long long)(K - 1) * dil * P;
  const bool lds = stride == 1 && halo <= 256 && !no_lds && (long long)Tin * P < (1ll << 31);
  if (lds) {
    // short rows: several channel sub-rows per workgroup
    const int TU = U < 256 ? U : 256;
    const int R = 256 / TU;
    const int ntl = vcv_cdiv(U, TU);
    splits = 1;
    if (out_act == VCV_ACT_NONE && !vcv_get_deterministic())
      while (splits < 64 && (long long)ntl * B * splits < 1024 && C / (splits * 2) >= 8 * R) splits *= 2;
    const int cperl = vcv_cdiv(C, splits);
    const int W = TU + (int)halo;
    int cch = (40 * 1024 / 4 - 256) / (W + K);
    if (cch > cperl) cch = cperl;
    cch = (cch + 7) & ~7;
    const size_t smem = sizeof(float) * ((size_t)cch * (W + K) + 256);
    if (splits > 1) {
      const size_t n = (size_t)B * U;
      hipLaunchKernelGGL(fill_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, bias, n);
    }
    hipLaunchKernelGGL(conv_m1_lds_kernel, dim3(ntl, B, vcv_cdiv(C, cperl)), dim3(256), smem, st, x, w, bias, y, C, Tin * P,
                       U, K, dil * P, pad * P, in_leaky, out_act, slope, cperl, cch, TU, R, W, splits > 1 ? 1 : 0,
                       ntl == 1 && Tout == Tin + 2 * pad - dil * (K - 1) ? 1 : 0);
    return vcv_check_launch();
  }
  const int cper = vcv_cdiv(C, splits);
  if (splits > 1) {
    const size_t n = (size_t)B * U;
    hipLaunchKernelGGL(fill_bias_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, y, bias, n);
  }
  hipLaunchKernelGGL(conv_m1_fwd_kernel, dim3(nt, B, splits), dim3(256), 0, st, x, w, bias, y, C, Tin, Tout, P, K,
                     stride, dil, pad, in_leaky, out_act, slope, cper, splits > 1 ? 1 : 0);
  return vcv_check_launch();
}

extern "C" int vcv_thin_wgrad(const float* a, const float* bsh, const float* aaux, const float* baux, float* dw,
                              int B, int M, int C, int Ta, int Tb, int P, int K, int s, int d, int off, int a_tf,
                              int b_tf, float slope, float alpha, void* stream) {
  if (!a || !bsh || !dw || B <= 0 || M <= 0 || C <= 0 || Ta <= 0 || Tb <= 0 || P <= 0 || K <= 0 || K > KMAX)
    return VCV_EINVAL;
  if ((a_tf >= VCV_TF_DLEAKY && !aaux) || (b_tf >= VCV_TF_DLEAKY && !baux)) return VCV_EINVAL;
  if ((long long)Ta * P >= (1ll << 31) || (long long)Tb * P >= (1ll << 31)) return VCV_EINVAL;
  const bool no_pairs = !vcv_tuning().c1_wgrad_pairs;
  // (stride 1 only: with a stride the x index is not linear in the flattened position and the two-level loop it
  // needs measured slower than the row-per-workgroup kernel on the period discriminators' 1 -> 32 k5 s3 layers)
  if (C == 1 && s == 1 && M * K <= 256 && a_tf == VCV_TF_NONE && b_tf == VCV_TF_NONE && !vcv_get_deterministic() &&
      !no_pairs && P <= 256) {
    // row tile of <= 256 positions, chunk of ~1024 positions per workgroup
    int tq = 256 / P;
    if (tq > Ta) tq = Ta;
    int qper = tq * (1024 / (tq * P) > 0 ? 1024 / (tq * P) : 1);
    if (qper > Ta) qper = vcv_cdiv(Ta, tq) * tq;
    const int xw = ((qper - 1) * s + (K - 1) * d + 1) * P;
    int TUP = tq * P;
    if (TUP % 32 == 0) TUP += 1;
    const size_t smem = sizeof(float) * ((size_t)xw + (size_t)M * TUP);
    if (smem <= 60 * 1024) {
      hipLaunchKernelGGL(c1_wgrad_pairs_kernel, dim3(vcv_cdiv(Ta, qper), B), dim3(256), smem, (hipStream_t)stream, a, bsh, dw,
                         M, Ta, Tb, P, K, s, d, off, alpha, qper, tq, TUP, xw);
      return vcv_check_launch();
    }
  }
  int splits = 1;
  while (splits < B && (long long)M * C * splits < 4096) splits *= 2;  // short serial batch loops: latency-bound
  if (splits > B) splits = B;
  const bool det = vcv_get_deterministic() != 0;  // one workgroup per (m, c): a single writer per weight
  if (det) splits = 1;
  const int bper = vcv_cdiv(B, splits);
  // long rows: also split the positions so that the grid has a few thousand workgroups of >= 1024 positions
  const int U = Ta * P;
  const int wg_target = vcv_tuning().thin_wgrad_wgs > 0 ? vcv_tuning().thin_wgrad_wgs : 2048;
  long long usplit = wg_target / ((long long)M * C * vcv_cdiv(B, bper));
  if (usplit > U / 1024) usplit = U / 1024;
  if (usplit < 1 || det) usplit = 1;
  const int uper = (vcv_cdiv(U, (int)usplit) + 255) & ~255;
  const dim3 grid(M * C, vcv_cdiv(B, bper), vcv_cdiv(U, uper));
#define VCV_THIN_WGRAD(PLAIN, KT)                                                                                        \
  hipLaunchKernelGGL((thin_wgrad_kernel<PLAIN, KT>), grid, dim3(256), 0, (hipStream_t)stream, a, bsh, aaux, baux, dw, B, M, \
                     C, Ta, Tb, P, K, s, d, off, a_tf, b_tf, slope, alpha, bper, uper)
  if (a_tf == VCV_TF_NONE && b_tf == VCV_TF_NONE) {
    switch (K) {
      case 3: VCV_THIN_WGRAD(true, 3); break;
      case 5: VCV_THIN_WGRAD(true, 5); break;
      case 7: VCV_THIN_WGRAD(true, 7); break;
      default: VCV_THIN_WGRAD(true, 0); break;
    }
  } else {
    VCV_THIN_WGRAD(false, 0);
  }
#undef VCV_THIN_WGRAD
  return vcv_check_launch();
}

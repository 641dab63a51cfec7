// conv_grouped.hip -- the grouped k=41 stride-4 convolutions of DiscriminatorS
// (discriminator.py:55-58: 16->64 g4, 64->256 g16, 256->1024 g64, 1024->1024 g256; 4 input
// channels per group, 16 or 4 output channels per group).
//
// With 4x(4|16) channels per group a 32x32 MFMA tile would be 75-88 % padding.  The 16-channel groups' forward
// and weight gradient run on v_mfma_f32_16x16x4_f32 (one group = one 16-row tile, K = the 4 input channels: no
// padding; grouped_fwd_mfma_kernel / grouped_wgrad_mfma_kernel below); the 4-channel groups and every data
// gradient (4 output rows per group) run as direct fp32 FMA kernels: a workgroup owns one (batch element, group, time tile); the group's
// input span and its (tiny) weight block sit in LDS; each lane computes a few output times for
// ALL channels of the group from 16-byte LDS reads (conflict-free x windows, broadcast weights).
//   forward   y[b,g*Mg+m,t]   = act(bias + sum_{ci,k} w[g*Mg+m,ci,k] * x[b,g*4+ci,4t+k-20])
//   dgrad     dx[b,g*4+ci,u]  = sum_{m,k == (u+20) mod 4} w[g*Mg+m,ci,k] * dye[b,g*Mg+m,(u+20-k)/4]
//   wgrad     dw[g*Mg+m,ci,k] += sum_{b,t} dye[b,g*Mg+m,t] * x[b,g*4+ci,4t+k-20]
// with dye = dy * leaky'(y) (the activation derivative is applied while staging).
#include "common.h"

namespace {

constexpr int CG = 4, K = 41, S = 4, PAD = 20, KP = 44;  // taps padded to a multiple of 4

// Stage N (compile-time) elements with the 256 lanes of a workgroup: LB loads are issued before the first of them
// is consumed.  A plain `for (i = tid; i < N; i += 256) lds[i] = src[...]` loop is compiled to one load, one wait,
// one LDS write per iteration -- 17 serial memory latencies per 256-time stage, which was 3/4 of these kernels' time.
template <int N, int LB, class Src, class Dst>
__device__ __forceinline__ void stage_n(int tid, Src src, Dst dst) {
  constexpr int IT = (N + 255) / 256;
#pragma unroll
  for (int i0 = 0; i0 < IT; i0 += LB) {
    float v[LB];
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int i = tid + (i0 + l) * 256;
      v[l] = (i0 + l < IT && i < N) ? src(i) : 0.f;
    }
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int i = tid + (i0 + l) * 256;
      if (i0 + l < IT && i < N) dst(i, v[l]);
    }
  }
}

// the same with a second operand (the activation output of a fused derivative), loaded in its own batch when wanted
template <int N, int LB, class Src, class Aux, class Dst>
__device__ __forceinline__ void stage_n2(int tid, bool want_aux, Src src, Aux aux, Dst dst) {
  constexpr int IT = (N + 255) / 256;
#pragma unroll
  for (int i0 = 0; i0 < IT; i0 += LB) {
    float v[LB], a[LB];
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int i = tid + (i0 + l) * 256;
      v[l] = (i0 + l < IT && i < N) ? src(i) : 0.f;
      a[l] = 1.f;
    }
    if (want_aux) {
#pragma unroll
      for (int l = 0; l < LB; ++l) {
        const int i = tid + (i0 + l) * 256;
        if (i0 + l < IT && i < N) a[l] = aux(i);
      }
    }
#pragma unroll
    for (int l = 0; l < LB; ++l) {
      const int i = tid + (i0 + l) * 256;
      if (i0 + l < IT && i < N) dst(i, v[l], a[l]);
    }
  }
}

// ---- forward: block = 256 lanes = GB groups x TT output times (TT = 256 / GB: short pooled scales would leave
// three quarters of a 256-time tile idle); lane = one output time of one group, all Mg outputs ----
template <int MG, int TT>
__global__ void __launch_bounds__(256)
grouped_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ y, int G, int Tin, int Tout, int act, float slope) {
  constexpr int GB = 256 / TT;
  constexpr int SPAN = S * TT + KP;  // input samples per channel needed by the tile
  __shared__ __attribute__((aligned(16))) float xs[GB][CG][SPAN];
  __shared__ __attribute__((aligned(16))) float ws[GB][CG][KP][MG];  // [ci][k][m]: m contiguous -> b128 broadcast reads
  const int tid = threadIdx.x;
  const int gl = tid / TT, tl = tid - gl * TT;
  const int t0 = blockIdx.x * TT, g0 = blockIdx.y * GB, b = blockIdx.z;
  const int in0 = t0 * S - PAD;
  stage_n<GB * CG * SPAN, 10>(tid,
      [&](int i) {
        const int gg = i / (CG * SPAN), r = i - gg * (CG * SPAN);
        const int ci = r / SPAN, ti = in0 + r - ci * SPAN;
        const bool ok = g0 + gg < G && ti >= 0 && ti < Tin;
        return ok ? x[((size_t)b * G * CG + (size_t)(g0 + gg) * CG + ci) * Tin + ti] : 0.f;
      },
      [&](int i, float v) { (&xs[0][0][0])[i] = v; });
  stage_n<GB * CG * KP * MG, 11>(tid,
      [&](int i) {
        const int gg = i / (CG * KP * MG), r = i - gg * (CG * KP * MG);
        const int m = r % MG, k = (r / MG) % KP, ci = r / (MG * KP);
        return (k < K && g0 + gg < G) ? w[((size_t)((g0 + gg) * MG + m) * CG + ci) * K + k] : 0.f;
      },
      [&](int i, float v) { (&ws[0][0][0][0])[i] = v; });
  __syncthreads();
  float acc[MG];
#pragma unroll
  for (int m = 0; m < MG; ++m) acc[m] = 0.f;
#pragma unroll 1
  for (int ci = 0; ci < CG; ++ci) {
#pragma unroll 1
    for (int k4 = 0; k4 < KP; k4 += 4) {
      const f32x4 xv = *(const f32x4*)&xs[gl][ci][S * tl + k4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float xk = xv[kk];
#pragma unroll
        for (int m4 = 0; m4 < MG; m4 += 4) {
          const f32x4 wv = *(const f32x4*)&ws[gl][ci][k4 + kk][m4];
          acc[m4 + 0] += wv[0] * xk; acc[m4 + 1] += wv[1] * xk;
          acc[m4 + 2] += wv[2] * xk; acc[m4 + 3] += wv[3] * xk;
        }
      }
    }
  }
  const int t = t0 + tl, g = g0 + gl;
  if (t < Tout && g < G) {
#pragma unroll
    for (int m = 0; m < MG; ++m) {
      const int mg = g * MG + m;
      float v = acc[m] + (bias ? bias[mg] : 0.f);
      y[((size_t)b * G * MG + mg) * Tout + t] = vcv_act(v, act, slope);
    }
  }
}

// ---- dgrad: lane = one quad of input times u = 4q..4q+3 (the 4 residues), all 4 input channels ----
// dx[ci][4q+r] = sum_m sum_j w[m][ci][r + 4j] * dye[m][q + 5 - j]   (j = 0..10; k = r+4j <= 40)
template <int MG, int QT>
__global__ void __launch_bounds__(256)
grouped_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yaux, const float* __restrict__ w,
                     float* __restrict__ dx, int G, int Tin, int Tout, int dtf, float slope) {
  constexpr int GB = 256 / QT;     // groups per block (QT quads = 4*QT input times each)
  constexpr int DSPAN = QT + 12;   // dy positions q-5 .. q+QT-1+5 (+ pad)
  __shared__ float ds[GB][MG][DSPAN];
  __shared__ __attribute__((aligned(16))) float ws[GB][MG][KP][CG];  // [m][k][ci]
  const int tid = threadIdx.x;
  const int gl = tid / QT, ql = tid - gl * QT;
  const int q0 = blockIdx.x * QT, g0 = blockIdx.y * GB, b = blockIdx.z;
  {
    auto at = [&](int i, bool& ok) {
      const int gg = i / (MG * DSPAN), r = i - gg * (MG * DSPAN);
      const int m = r / DSPAN, t = q0 - 5 + r - m * DSPAN;
      ok = t >= 0 && t < Tout && g0 + gg < G;
      return ((size_t)b * G * MG + (size_t)(g0 + gg) * MG + m) * Tout + t;
    };
    stage_n2<GB * MG * DSPAN, 8>(tid, dtf == VCV_TF_DLEAKY,
        [&](int i) { bool ok; const size_t gi = at(i, ok); return ok ? dy[gi] : 0.f; },
        [&](int i) { bool ok; const size_t gi = at(i, ok); return ok ? yaux[gi] : 1.f; },
        [&](int i, float v, float a) { (&ds[0][0][0])[i] = v * vcv_dleaky(a, slope); });
  }
  stage_n<GB * MG * KP * CG, 11>(tid,
      [&](int i) {
        const int gg = i / (MG * KP * CG), r = i - gg * (MG * KP * CG);
        const int ci = r % CG, k = (r / CG) % KP, m = r / (CG * KP);
        return (k < K && g0 + gg < G) ? w[((size_t)((g0 + gg) * MG + m) * CG + ci) * K + k] : 0.f;
      },
      [&](int i, float v) { (&ws[0][0][0][0])[i] = v; });
  __syncthreads();
  float acc[4][CG];  // [r][ci]
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int ci = 0; ci < CG; ++ci) acc[r][ci] = 0.f;
  // dy index for (q, j): t = q + 5 - j  -> ds column (q - q0) + 10 - j
#pragma unroll 2
  for (int m = 0; m < MG; ++m) {
#pragma unroll
    for (int j = 0; j < 11; ++j) {
      const float dv = ds[gl][m][ql + 10 - j];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4 wv = *(const f32x4*)&ws[gl][m][r + 4 * j][0];  // k = r + 4j (k > 40 rows are zero)
        acc[r][0] += wv[0] * dv; acc[r][1] += wv[1] * dv; acc[r][2] += wv[2] * dv; acc[r][3] += wv[3] * dv;
      }
    }
  }
  // u + 20 = 4q' + r with q' = q + 5  ->  u = 4(q + 5) + r - 20 = 4q + r
  const int q = q0 + ql, g = g0 + gl;
  if (g >= G) return;
  float* dxb = dx + ((size_t)b * G * CG + (size_t)g * CG) * Tin;
#pragma unroll
  for (int ci = 0; ci < CG; ++ci)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int u = 4 * q + r;
      if (u < Tin) dxb[(size_t)ci * Tin + u] = acc[r][ci];
    }
}

// ---- wgrad: block = (group, batch element, time chunk); lane = (m, ci, k4) owning 4 taps ----
template <int MG>
__global__ void __launch_bounds__(256)
grouped_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yaux, const float* __restrict__ x,
                     float* __restrict__ dw, int G, int Tin, int Tout, int dtf, float slope, int tchunk, int bper, int B) {
  constexpr int TT = 128;  // output times per stage
  constexpr int SPAN = S * TT + KP;
  constexpr int NOUT = MG * CG * (KP / 4);  // lanes with work: MG*4*11
  __shared__ __attribute__((aligned(16))) float xs[CG][SPAN];
  __shared__ float ds[MG][TT];
  const int tid = threadIdx.x;
  const int g = blockIdx.x;
  const int tlo = blockIdx.y * tchunk;
  int thi = tlo + tchunk;
  if (thi > Tout) thi = Tout;
  // each thread may own up to ceil(NOUT/256) (m, ci, k4) triples
  constexpr int NOWN = (NOUT + 255) / 256;
  float acc[NOWN][4];
#pragma unroll
  for (int o = 0; o < NOWN; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;
  // blockIdx.z takes bper batch elements: the partial sums stay in registers across them (one set of atomics)
  const int b_lo = blockIdx.z * bper, b_hi = b_lo + bper < B ? b_lo + bper : B;
  for (int b = b_lo; b < b_hi; ++b) {
  const float* xb = x + ((size_t)b * G * CG + (size_t)g * CG) * Tin;
  const size_t ybase = ((size_t)b * G * MG + (size_t)g * MG) * Tout;
  for (int t0 = tlo; t0 < thi; t0 += TT) {
    __syncthreads();
    const int in0 = t0 * S - PAD;
    stage_n<CG * SPAN, 10>(tid,
        [&](int i) {
          const int ci = i / SPAN, ti = in0 + i - ci * SPAN;
          return (ti >= 0 && ti < Tin) ? xb[(size_t)ci * Tin + ti] : 0.f;
        },
        [&](int i, float v) { (&xs[0][0])[i] = v; });
    stage_n2<MG * TT, 8>(tid, dtf == VCV_TF_DLEAKY,
        [&](int i) {
          const int m = i / TT, t = t0 + i - m * TT;
          return t < thi ? dy[ybase + (size_t)m * Tout + t] : 0.f;
        },
        [&](int i) {
          const int m = i / TT, t = t0 + i - m * TT;
          return t < thi ? yaux[ybase + (size_t)m * Tout + t] : 1.f;
        },
        [&](int i, float v, float a) { ds[i / TT][i % TT] = v * vcv_dleaky(a, slope); });
    __syncthreads();
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
      const int id = tid + o * 256;
      if (id < NOUT) {
        const int k4 = (id % (KP / 4)) * 4, ci = (id / (KP / 4)) % CG, m = id / (CG * (KP / 4));
#pragma unroll 4
        for (int j = 0; j < TT; ++j) {
          const float dv = ds[m][j];
          const f32x4 xv = *(const f32x4*)&xs[ci][S * j + k4];
          acc[o][0] += dv * xv[0]; acc[o][1] += dv * xv[1]; acc[o][2] += dv * xv[2]; acc[o][3] += dv * xv[3];
        }
      }
    }
  }
  }
#pragma unroll
  for (int o = 0; o < NOWN; ++o) {
    const int id = tid + o * 256;
    if (id < NOUT) {
      const int k4 = (id % (KP / 4)) * 4, ci = (id / (KP / 4)) % CG, m = id / (CG * (KP / 4));
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (k4 + e < K) unsafeAtomicAdd(dw + ((size_t)(g * MG + m) * CG + ci) * K + k4 + e, acc[o][e]);
    }
  }
}

}  // namespace

// x [B, G*4, Tin], w [G*Mg, 4, 41], y [B, G*Mg, Tout], Tout = (Tin + 40 - 41)/4 + 1; Mg in {4, 16}
namespace {
// ---- forward on the matrix cores (16 output channels per group = one 16x16 tile) -----------------------------
// v_mfma_f32_16x16x4_f32 per tap: A[m][ci] = w[g*16+m][ci][tap], B[ci][n] = x[g*4+ci][4*(t0+n) + tap - 20], so the
// K dimension of the instruction is exactly the group's 4 input channels and nothing is padded.  The input span
// is staged de-interleaved by position mod 4 (xs[ci][phase][q]): for one tap all 16 columns read consecutive
// floats (conflict-free), the weights are staged [tap][ci][m] (a 64-float row per instruction).
// Workgroup = 4 waves = 4 x 64 output times of one (batch, group); grid (Tout/256, G, B).
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256)
grouped_fwd_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                        float* __restrict__ y, int G, int Tin, int Tout, int act, float slope) {
  constexpr int TT = 256, MG = 16;
  constexpr int QN = TT + 11;           // quarter-rate samples per phase: 4*(TT-1) + 40 < 4*QN
  constexpr int QP = QN + 1;            // phase pitch
  __shared__ float xs[CG][4][QP];
  __shared__ float ws[K][CG * MG + 1];  // [tap][ci*16 + m], odd pitch: the staging scatter below is conflict-free
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * TT, g = blockIdx.y, b = blockIdx.z;
  const float* xb = x + ((size_t)b * G * CG + (size_t)g * CG) * Tin;
  const int in0 = t0 * S - PAD;  // a multiple of 4
  stage_n<CG * 4 * QN, 9>(tid,
      [&](int i) {
        const int ci = i / (4 * QN), ti = in0 + i - ci * (4 * QN);  // offset inside the span: coalesced global reads
        return (ti >= 0 && ti < Tin) ? xb[(size_t)ci * Tin + ti] : 0.f;
      },
      [&](int i, float v) {
        const int ci = i / (4 * QN), j = i - ci * (4 * QN);
        xs[ci][j & 3][j >> 2] = v;
      });
  const float* wg = w + (size_t)g * MG * CG * K;  // the group's [m][ci][k] block is contiguous: coalesced reads
  stage_n<K * CG * MG, 11>(tid, [&](int i) { return wg[i]; },
      [&](int i, float v) {
        const int k = i % K, ci = (i / K) % CG, m = i / (K * CG);
        ws[k][ci * MG + m] = v;
      });
  __syncthreads();
  const int n = lane & 15, kq = lane >> 4;  // column / k index of this lane's A and B elements
  f32x4_t acc[4];
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) acc[tl] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int tw = wave * 64;  // first output time of this wave inside the tile
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const float a = ws[k][lane];                  // A[m = n][ci = kq]: lane = kq*16 + n
    const float* xr = &xs[kq][k & 3][(k >> 2) + tw + n];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xr[tl * 16], acc[tl], 0, 0, 0);
  }
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const int t = t0 + tw + tl * 16 + n;
    if (t >= Tout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = kq * 4 + r;
      float v = acc[tl][r] + (bias ? bias[g * MG + m] : 0.f);
      y[((size_t)b * G * MG + (size_t)g * MG + m) * Tout + t] = vcv_act(v, act, slope);
    }
  }
}
}  // namespace

namespace {
// ---- forward, bf16 operands (bf16 mode of the library: operands rounded on their way into the matrix cores, fp32
// accumulate).  v_mfma_f32_16x16x16_bf16 takes K = 16 = (4 taps) x (the group's 4 input channels): lane quarter kq holds
// tap 4 tg + kq and, as its four bf16 elements, the four channels -- 11 MFMA steps per 16 x 16 tile instead of 41 with the
// fp32 shape (the fp32-input kernel above is at 69 % of the fp32 MFMA peak: in bf16 mode these three layers were the one
// GEMM-shaped launch family still on the fp32 pipe, 5.7 ms of the 84 ms configs[2] step).  With the taps grouped in fours the
// stride-4 input index 4 (t0 + n) + 4 tg + kq has residue kq: the input span is staged as [residue][q] items of four
// channels (8 bytes of bf16: one ds_read_b64 per B fragment), the weights as [tap group][kq][m] items of four channels.
typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4_t pack_bf16x4(float a, float b, float c, float d) {
  typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
  bf16x4_t v;
  v[0] = (__bf16)a, v[1] = (__bf16)b, v[2] = (__bf16)c, v[3] = (__bf16)d;
  return __builtin_bit_cast(s16x4_t, v);
}

__global__ void __launch_bounds__(256)
grouped_fwd_mfma_bf16_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                             float* __restrict__ y, int G, int Tin, int Tout, int act, float slope) {
  constexpr int TT = 256, MG = 16, NTG = KP / 4;  // 11 tap groups
  constexpr int QN = TT + 11;                     // quarter-rate samples per residue
  __shared__ s16x4_t xs[4][QN + 1];               // [residue][q]: channels 0..3 of input 4 q + residue
  __shared__ s16x4_t wsb[NTG][4][MG];             // [tap group][kq][m]: channels 0..3 of tap 4 tg + kq (taps >= 41: zero)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t0 = blockIdx.x * TT, g = blockIdx.y, b = blockIdx.z;
  const float* xb = x + ((size_t)b * G * CG + (size_t)g * CG) * Tin;
  const int in0 = t0 * S - PAD;  // a multiple of 4
  // input span: thread = position, the four channel rows loaded together (coalesced per row), five positions in flight
  for (int j0 = tid; j0 < 4 * QN; j0 += 256 * 5) {
    float v[5][4];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int j = j0 + u * 256, ti = in0 + j;
      const bool ok = j < 4 * QN && ti >= 0 && ti < Tin;
#pragma unroll
      for (int ci = 0; ci < 4; ++ci) v[u][ci] = ok ? xb[(size_t)ci * Tin + ti] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int j = j0 + u * 256;
      if (j < 4 * QN) xs[j & 3][j >> 2] = pack_bf16x4(v[u][0], v[u][1], v[u][2], v[u][3]);
    }
  }
  // weights: item (tap group, kq, m) = w[g*16 + m][0..3][4 tg + kq]
  const float* wg = w + (size_t)g * MG * CG * K;
  for (int i = tid; i < NTG * 4 * MG; i += 256) {
    const int m = i % MG, kq = (i / MG) & 3, tg = i / (4 * MG);
    const int k = 4 * tg + kq;
    float v[4];
#pragma unroll
    for (int ci = 0; ci < 4; ++ci) v[ci] = k < K ? wg[((size_t)m * CG + ci) * K + k] : 0.f;
    wsb[tg][kq][m] = pack_bf16x4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  const int n = lane & 15, kq = lane >> 4;
  f32x4_t acc[4];
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) acc[tl] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int tw = wave * 64;
#pragma unroll
  for (int tg = 0; tg < NTG; ++tg) {
    const s16x4_t a = wsb[tg][kq][n];  // A[m = n][k = (kq, ci)]
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)     // B[k = (kq, ci)][column]: input 4 (tw + 16 tl + n) + 4 tg + kq
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, xs[kq][tw + tl * 16 + n + tg], acc[tl], 0, 0, 0);
  }
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const int t = t0 + tw + tl * 16 + n;
    if (t >= Tout) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = kq * 4 + r;
      float v = acc[tl][r] + (bias ? bias[g * MG + m] : 0.f);
      y[((size_t)b * G * MG + (size_t)g * MG + m) * Tout + t] = vcv_act(v, act, slope);
    }
  }
}
}  // namespace

// bf16-operand form of vcv_grouped41_fwd for the 16-channel groups (Mg == 16; other group widths: VCV_EINVAL, the caller
// keeps vcv_grouped41_fwd)
extern "C" int vcv_grouped41_fwd_bf16(const float* x, const float* w, const float* bias, float* y, int B, int G, int Mg,
                                      int Tin, int Tout, int out_act, float slope, void* stream) {
  if (!x || !w || !y || B <= 0 || G <= 0 || Tin <= 0 || Tout <= 0 || Mg != 16) return VCV_EINVAL;
  hipLaunchKernelGGL(grouped_fwd_mfma_bf16_kernel, dim3(vcv_cdiv(Tout, 256), G, B), dim3(256), 0, (hipStream_t)stream, x, w, bias,
                     y, G, Tin, Tout, out_act, slope);
  return vcv_check_launch();
}

extern "C" int vcv_grouped41_fwd(const float* x, const float* w, const float* bias, float* y, int B, int G, int Mg,
                                 int Tin, int Tout, int out_act, float slope, void* stream) {
  if (!x || !w || !y || B <= 0 || G <= 0 || Tin <= 0 || Tout <= 0 || (Mg != 4 && Mg != 16)) return VCV_EINVAL;
  dim3 grid(vcv_cdiv(Tout, 256), G, B);
  if (Mg == 16) hipLaunchKernelGGL(grouped_fwd_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, G, Tin, Tout, out_act, slope);
  else if (Tout <= 64)
    hipLaunchKernelGGL((grouped_fwd_kernel<4, 64>), dim3(vcv_cdiv(Tout, 64), vcv_cdiv(G, 4), B), dim3(256), 0, (hipStream_t)stream, x,
                       w, bias, y, G, Tin, Tout, out_act, slope);
  else if (Tout <= 128)
    hipLaunchKernelGGL((grouped_fwd_kernel<4, 128>), dim3(vcv_cdiv(Tout, 128), vcv_cdiv(G, 2), B), dim3(256), 0, (hipStream_t)stream,
                       x, w, bias, y, G, Tin, Tout, out_act, slope);
  else
    hipLaunchKernelGGL((grouped_fwd_kernel<4, 256>), grid, dim3(256), 0, (hipStream_t)stream, x, w, bias, y, G, Tin, Tout, out_act,
                       slope);
  return vcv_check_launch();
}

namespace {
// ---- data gradient on the matrix cores (16 output channels per group) -----------------------------------------
// dx[ci][4q+r] = sum_j sum_m w[m][ci][r+4j] * dye[m][q+5-j]: with rows = (ci, r) -- the 4 input channels x the 4
// residues of the stride -- this is a 16-row GEMM whose shift (5 - j) does not depend on the row, so each of the
// 11 tap groups is four v_mfma_f32_16x16x4_f32 (K = the 16 output channels in chunks of 4).  A lane ends up with
// the four residues of one (ci, q): one 16-byte store, 256 B per 16 lanes.
__global__ void __launch_bounds__(256)
grouped_dgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ yaux, const float* __restrict__ w,
                          float* __restrict__ dx, int G, int Tin, int Tout, int dtf, float slope) {
  constexpr int MG = 16, QT = 256, NJ = 11;
  constexpr int DSP = 272;  // >= QT + 10 + 1, and == 16 (mod 32): the two k-halves of a B fragment hit disjoint banks
  __shared__ float ds[MG][DSP];
  __shared__ float wsd[NJ * 4][64];  // [(j, m chunk)][kq*16 + ci*4 + r]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * QT, g = blockIdx.y, b = blockIdx.z;
  const size_t ybase = ((size_t)b * G * MG + (size_t)g * MG) * Tout;
  {
    auto at = [&](int i, bool& ok) {
      const int m = i / DSP, t = q0 - 5 + i - m * DSP;
      ok = t >= 0 && t < Tout;
      return ybase + (size_t)m * Tout + t;
    };
    stage_n2<MG * DSP, 9>(tid, dtf == VCV_TF_DLEAKY,
        [&](int i) { bool ok; const size_t gi = at(i, ok); return ok ? dy[gi] : 0.f; },
        [&](int i) { bool ok; const size_t gi = at(i, ok); return ok ? yaux[gi] : 1.f; },
        [&](int i, float v, float a) { (&ds[0][0])[i] = v * vcv_dleaky(a, slope); });
  }
  for (int i = tid; i < NJ * 4 * 64; i += 256) (&wsd[0][0])[i] = 0.f;
  __syncthreads();
  const float* wg = w + (size_t)g * MG * CG * K;
  stage_n<MG * CG * K, 11>(tid, [&](int i) { return wg[i]; },
      [&](int i, float v) {
        const int k = i % K, ci = (i / K) % CG, m = i / (K * CG);
        wsd[(k >> 2) * 4 + (m >> 2)][(m & 3) * 16 + ci * 4 + (k & 3)] = v;
      });
  __syncthreads();
  const int n = lane & 15, kq = lane >> 4;
  f32x4_t acc[4];
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) acc[tl] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int qw = wave * 64;
#pragma unroll 1
  for (int j = 0; j < NJ; ++j) {
#pragma unroll
    for (int mc = 0; mc < 4; ++mc) {
      const float a = wsd[j * 4 + mc][lane];                     // A[row = (ci, r)][m = mc*4 + kq]
      const float* dr = &ds[mc * 4 + kq][qw + n + 10 - j];       // B[m][q]: dye[m][q + 5 - j]
#pragma unroll
      for (int tl = 0; tl < 4; ++tl) acc[tl] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, dr[tl * 16], acc[tl], 0, 0, 0);
    }
  }
  // lane holds rows 4*kq .. 4*kq+3 = (ci = kq, r = 0..3) of column q: four consecutive input times
  float* dxr = dx + ((size_t)b * G * CG + (size_t)g * CG + kq) * Tin;
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const int q = q0 + qw + tl * 16 + n;
    const int u = 4 * q;
    if (u + 3 < Tin && (Tin & 3) == 0) {
      *reinterpret_cast<f32x4_t*>(dxr + u) = acc[tl];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (u + r < Tin) dxr[u + r] = acc[tl][r];
    }
  }
}
}  // namespace

namespace {
// ---- data gradient, bf16 operands: rows = (ci, r) as above, K of v_mfma_f32_16x16x16_bf16 = ALL 16 output channels of the
// group, so each of the 11 tap groups is ONE MFMA per 16 x 16 tile (four with the fp32 shape).  dye = dy * leaky'(y) is
// staged as [m / 4][position] items of four channels (8 bytes of bf16), the weights as [tap group][kq][row] items.
__global__ void __launch_bounds__(256)
grouped_dgrad_mfma_bf16_kernel(const float* __restrict__ dy, const float* __restrict__ yaux, const float* __restrict__ w,
                               float* __restrict__ dx, int G, int Tin, int Tout, int dtf, float slope) {
  constexpr int MG = 16, QT = 256, NJ = 11;
  constexpr int DSP = QT + 10 + 1;
  __shared__ s16x4_t dsb[4][DSP];      // [m chunk kq][position]: channels 4 kq .. 4 kq + 3 of dye at q0 - 5 + position
  __shared__ s16x4_t wsd[NJ][4][16];   // [tap group j][kq][row = ci * 4 + r]: channels 4 kq .. + 3 of w[m][ci][r + 4 j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q0 = blockIdx.x * QT, g = blockIdx.y, b = blockIdx.z;
  const size_t ybase = ((size_t)b * G * MG + (size_t)g * MG) * Tout;
  // dye: thread = position, the 16 channel rows loaded together (coalesced per row), two positions in flight
  for (int p0 = tid; p0 < DSP; p0 += 256 * 2) {
    float v[2][MG], a[2][MG];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pp = p0 + u * 256, t = q0 - 5 + pp;
      const bool ok = pp < DSP && t >= 0 && t < Tout;
#pragma unroll
      for (int m = 0; m < MG; ++m) {
        v[u][m] = ok ? dy[ybase + (size_t)m * Tout + t] : 0.f;
        a[u][m] = (ok && dtf == VCV_TF_DLEAKY) ? yaux[ybase + (size_t)m * Tout + t] : 1.f;
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int pp = p0 + u * 256;
      if (pp < DSP) {
#pragma unroll
        for (int mc = 0; mc < 4; ++mc)
          dsb[mc][pp] = pack_bf16x4(v[u][4 * mc] * vcv_dleaky(a[u][4 * mc], slope), v[u][4 * mc + 1] * vcv_dleaky(a[u][4 * mc + 1], slope),
                                    v[u][4 * mc + 2] * vcv_dleaky(a[u][4 * mc + 2], slope), v[u][4 * mc + 3] * vcv_dleaky(a[u][4 * mc + 3], slope));
      }
    }
  }
  const float* wg = w + (size_t)g * MG * CG * K;
  for (int i = tid; i < NJ * 4 * 16; i += 256) {
    const int row = i & 15, kq = (i >> 4) & 3, j = i >> 6;
    const int ci = row >> 2, k = (row & 3) + 4 * j;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = k < K ? wg[((size_t)(4 * kq + e) * CG + ci) * K + k] : 0.f;
    wsd[j][kq][row] = pack_bf16x4(v[0], v[1], v[2], v[3]);
  }
  __syncthreads();
  const int n = lane & 15, kq = lane >> 4;
  f32x4_t acc[4];
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) acc[tl] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  const int qw = wave * 64;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const s16x4_t a = wsd[j][kq][n];  // A[row = n][k = m = 4 kq + e]
#pragma unroll
    for (int tl = 0; tl < 4; ++tl)    // B[k = m][column q]: dye[m][q + 5 - j]
      acc[tl] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, dsb[kq][qw + tl * 16 + n + 10 - j], acc[tl], 0, 0, 0);
  }
  float* dxr = dx + ((size_t)b * G * CG + (size_t)g * CG + kq) * Tin;
#pragma unroll
  for (int tl = 0; tl < 4; ++tl) {
    const int q = q0 + qw + tl * 16 + n;
    const int u = 4 * q;
    if (u + 3 < Tin && (Tin & 3) == 0) {
      *reinterpret_cast<f32x4_t*>(dxr + u) = acc[tl];
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (u + r < Tin) dxr[u + r] = acc[tl][r];
    }
  }
}
}  // namespace

extern "C" int vcv_grouped41_dgrad_bf16(const float* dy, const float* yaux, const float* w, float* dx, int B, int G, int Mg,
                                        int Tin, int Tout, int dtf, float slope, void* stream) {
  if (!dy || !w || !dx || B <= 0 || G <= 0 || Tin <= 0 || Tout <= 0 || Mg != 16) return VCV_EINVAL;
  if (dtf != VCV_TF_NONE && dtf != VCV_TF_DLEAKY) return VCV_EINVAL;
  if (dtf >= VCV_TF_DLEAKY && !yaux) return VCV_EINVAL;
  hipLaunchKernelGGL(grouped_dgrad_mfma_bf16_kernel, dim3(vcv_cdiv(vcv_cdiv(Tin, 4), 256), G, B), dim3(256), 0, (hipStream_t)stream, dy,
                     yaux, w, dx, G, Tin, Tout, dtf, slope);
  return vcv_check_launch();
}

extern "C" int vcv_grouped41_dgrad(const float* dy, const float* yaux, const float* w, float* dx, int B, int G, int Mg,
                                   int Tin, int Tout, int dtf, float slope, void* stream) {
  if (!dy || !w || !dx || B <= 0 || G <= 0 || Tin <= 0 || Tout <= 0 || (Mg != 4 && Mg != 16)) return VCV_EINVAL;
  if (dtf != VCV_TF_NONE && dtf != VCV_TF_DLEAKY) return VCV_EINVAL;  // (the staging applies the leaky-ReLU derivative only)
  if (dtf >= VCV_TF_DLEAKY && !yaux) return VCV_EINVAL;
  dim3 grid(vcv_cdiv(vcv_cdiv(Tin, 4), 256), G, B);
  if (Mg == 16) hipLaunchKernelGGL(grouped_dgrad_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, yaux, w, dx, G, Tin, Tout, dtf, slope);
  else {
    const int nq = vcv_cdiv(Tin, 4);
    if (nq <= 64)
      hipLaunchKernelGGL((grouped_dgrad_kernel<4, 64>), dim3(vcv_cdiv(nq, 64), vcv_cdiv(G, 4), B), dim3(256), 0, (hipStream_t)stream, dy,
                         yaux, w, dx, G, Tin, Tout, dtf, slope);
    else if (nq <= 128)
      hipLaunchKernelGGL((grouped_dgrad_kernel<4, 128>), dim3(vcv_cdiv(nq, 128), vcv_cdiv(G, 2), B), dim3(256), 0, (hipStream_t)stream,
                         dy, yaux, w, dx, G, Tin, Tout, dtf, slope);
    else
      hipLaunchKernelGGL((grouped_dgrad_kernel<4, 256>), grid, dim3(256), 0, (hipStream_t)stream, dy, yaux, w, dx, G, Tin, Tout, dtf,
                         slope);
  }
  return vcv_check_launch();
}

namespace {
// ---- weight gradient on the matrix cores (16 output channels per group) ---------------------------------------
// Per group a [16 x 164] result, reduction over (batch, time): v_mfma_f32_16x16x4_f32 with A[m][kk] = dye[m][t+kk]
// (4 consecutive output times), B[kk][(ci, k)] = x[ci][4*(t+kk) + k - 20]: 11 column tiles of 16 (164 -> 176,
// the padding columns are clamped reads and never stored).  Workgroup = (group, batch element, time chunk); its
// four waves take a quarter of each 256-time stage, add their partial tiles into one LDS tile and the workgroup
// leaves with one atomic per weight.
// BF: bf16 operands (bf16 mode): v_mfma_f32_16x16x16_bf16 with K = 16 consecutive output times per step (lane quarter kk
// holds times 4 kk .. 4 kk + 3) -- a quarter of the MFMA steps of the fp32 shape at the same number of LDS reads.
template <bool BF>
__global__ void __launch_bounds__(256)
grouped_wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ yaux, const float* __restrict__ x,
                          float* __restrict__ dw, int G, int Tin, int Tout, int dtf, float slope, int nstage, int uper,
                          int nunit) {
  constexpr int MG = 16, TT = 256, NT = 11, NW = CG * K;   // 164 weights per output channel
  constexpr int SPAN = S * TT + KP, DP = TT + 1;
  __shared__ float xs[CG][SPAN];
  __shared__ float ds[MG][DP];
  __shared__ float racc[MG][NT * 16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.x;
  // work units = (batch element, 256-time stage); this workgroup takes units [u_lo, u_hi) so that the whole grid
  // is ~512 workgroups and each leaves with one set of atomics
  const int u_lo = blockIdx.y * uper, u_hi = u_lo + uper < nunit ? u_lo + uper : nunit;
  const int j16 = lane & 15, kk = lane >> 4;
  int boff[NT];  // LDS offset of this lane's B element per column tile: xs[ci][4*kk + k]
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int n = nt * 16 + j16;
    if (n > NW - 1) n = NW - 1;
    const int ci = n / K, k = n - ci * K;
    boff[nt] = ci * SPAN + (BF ? 0 : S * kk) + k;
  }
  f32x4_t acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int i = tid; i < MG * NT * 16; i += 256) (&racc[0][0])[i] = 0.f;
  for (int unit = u_lo; unit < u_hi; ++unit) {
    const int b = unit / nstage, t0 = (unit - b * nstage) * TT;
    const float* xb = x + ((size_t)b * G * CG + (size_t)g * CG) * Tin;
    const size_t ybase = ((size_t)b * G * MG + (size_t)g * MG) * Tout;
    __syncthreads();
    const int in0 = t0 * S - PAD;
    stage_n<CG * SPAN, 9>(tid,
        [&](int i) {
          const int ci = i / SPAN, ti = in0 + i - ci * SPAN;
          return (ti >= 0 && ti < Tin) ? xb[(size_t)ci * Tin + ti] : 0.f;
        },
        [&](int i, float v) { (&xs[0][0])[i] = v; });
    stage_n2<MG * TT, 8>(tid, dtf == VCV_TF_DLEAKY,
        [&](int i) {
          const int m = i / TT, t = t0 + i - m * TT;
          return t < Tout ? dy[ybase + (size_t)m * Tout + t] : 0.f;
        },
        [&](int i) {
          const int m = i / TT, t = t0 + i - m * TT;
          return t < Tout ? yaux[ybase + (size_t)m * Tout + t] : 1.f;
        },
        [&](int i, float v, float a) { ds[i / TT][i % TT] = v * vcv_dleaky(a, slope); });
    __syncthreads();
    const float* xf = &xs[0][0];
    if constexpr (BF) {
#pragma unroll 1
      for (int tq = wave * 64; tq < wave * 64 + 64; tq += 16) {
        const float* ar = &ds[j16][tq + 4 * kk];   // A[m = j16][k = times tq + 4 kk + e]
        const s16x4_t a = pack_bf16x4(ar[0], ar[1], ar[2], ar[3]);
        const float* xq = xf + S * (tq + 4 * kk);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float* xp = xq + boff[nt];        // B[k = time][column (ci, tap)]: x[ci][4 (t + e) + tap - 20]
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, pack_bf16x4(xp[0], xp[S], xp[2 * S], xp[3 * S]), acc[nt], 0, 0, 0);
        }
      }
    } else {
#pragma unroll 2
      for (int tq = wave * 64; tq < wave * 64 + 64; tq += 4) {
        const float a = ds[j16][tq + kk];            // A[m = j16][kk]
        const float* xq = xf + S * tq;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xq[boff[nt]], acc[nt], 0, 0, 0);
      }
    }
  }
  // the four waves' partial tiles meet in LDS one wave after the other: a fixed order (LDS atomics would add them in
  // whatever order the waves arrive)
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) racc[kk * 4 + r][nt * 16 + j16] += acc[nt][r];
    }
  }
  __syncthreads();
  float* dwg = dw + (size_t)g * MG * NW;
  for (int i = tid; i < MG * NW; i += 256) {
    const int m = i / NW, n = i - m * NW;
    unsafeAtomicAdd(dwg + i, racc[m][n]);
  }
}
}  // namespace

static int grouped41_wgrad_impl(const float* dy, const float* yaux, const float* x, float* dw, int B, int G, int Mg,
                                int Tin, int Tout, int dtf, float slope, void* stream, bool bf) {
  if (!dy || !x || !dw || B <= 0 || G <= 0 || Tin <= 0 || Tout <= 0 || (Mg != 4 && Mg != 16)) return VCV_EINVAL;
  if (bf && Mg != 16) return VCV_EINVAL;
  if (dtf != VCV_TF_NONE && dtf != VCV_TF_DLEAKY) return VCV_EINVAL;  // (the staging applies the leaky-ReLU derivative only)
  if (dtf >= VCV_TF_DLEAKY && !yaux) return VCV_EINVAL;
  // time chunks so that the grid has >= ~1000 workgroups
  int nchunk = 1;
  if (Mg == 16) {
    const int nstage = vcv_cdiv(Tout, 256), nunit = B * nstage;
    int wg_per_group = 512 / G;
    if (wg_per_group < 1) wg_per_group = 1;
    if (wg_per_group > nunit) wg_per_group = nunit;
    if (vcv_get_deterministic()) wg_per_group = 1;  // one workgroup per group: a single writer per weight
    const int uper = vcv_cdiv(nunit, wg_per_group);
    if (bf)
      hipLaunchKernelGGL(grouped_wgrad_mfma_kernel<true>, dim3(G, vcv_cdiv(nunit, uper)), dim3(256), 0, (hipStream_t)stream, dy, yaux,
                         x, dw, G, Tin, Tout, dtf, slope, nstage, uper, nunit);
    else
      hipLaunchKernelGGL(grouped_wgrad_mfma_kernel<false>, dim3(G, vcv_cdiv(nunit, uper)), dim3(256), 0, (hipStream_t)stream, dy, yaux,
                         x, dw, G, Tin, Tout, dtf, slope, nstage, uper, nunit);
    return vcv_check_launch();
  }
  while ((long long)G * B * nchunk < 1024 && Tout / (nchunk * 2) >= 128) nchunk *= 2;
  const bool det = vcv_get_deterministic() != 0;
  if (det) nchunk = 1;
  const int tchunk = vcv_cdiv(vcv_cdiv(Tout, nchunk), 128) * 128;
  // batch elements per workgroup: as many as keep the grid at >= ~1024 workgroups (fewer atomics per weight)
  int bper = 1;
  while (bper * 2 <= B && (long long)G * vcv_cdiv(Tout, tchunk) * (B / (bper * 2)) >= 1024) bper *= 2;
  if (det) bper = B;
  dim3 grid(G, vcv_cdiv(Tout, tchunk), vcv_cdiv(B, bper));
  if (Mg == 16) hipLaunchKernelGGL(grouped_wgrad_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, dy, yaux, x, dw, G, Tin, Tout, dtf, slope, tchunk, bper, B);
  else hipLaunchKernelGGL(grouped_wgrad_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, dy, yaux, x, dw, G, Tin, Tout, dtf, slope, tchunk, bper, B);
  return vcv_check_launch();
}

extern "C" int vcv_grouped41_wgrad(const float* dy, const float* yaux, const float* x, float* dw, int B, int G, int Mg,
                                   int Tin, int Tout, int dtf, float slope, void* stream) {
  return grouped41_wgrad_impl(dy, yaux, x, dw, B, G, Mg, Tin, Tout, dtf, slope, stream, false);
}
// bf16-operand form for the 16-channel groups (Mg == 16, else VCV_EINVAL): K of the bf16 MFMA = 16 consecutive output times
extern "C" int vcv_grouped41_wgrad_bf16(const float* dy, const float* yaux, const float* x, float* dw, int B, int G, int Mg,
                                        int Tin, int Tout, int dtf, float slope, void* stream) {
  return grouped41_wgrad_impl(dy, yaux, x, dw, B, G, Mg, Tin, Tout, dtf, slope, stream, true);
}

// conv_gemm.hip -- implicit-GEMM convolution family on the gfx950 fp32 matrix cores.
//
// One kernel covers Conv1d / period Conv2d / ConvTranspose1d forward and all of their data
// gradients (see include/vcvits_hip.h for the parametrisation and the reference call sites).
//
// Mapping (MI355X-first, not a translation of a cuDNN/MIOpen call):
//   GEMM  M = output channels of a group, N = (q, p) positions of ONE batch element,
//         K = (reduction channel, tap).
//   A workgroup stages, per chunk of BKC reduction channels,
//     As[(c, j)][m]   the channel x tap weight tile, k-major so the MFMA A-fragment read
//                     (lane -> consecutive m) is one conflict-free ds_read_b32;
//     Xs[c][row*P+p]  ONE contiguous time span of the input per channel -- every tap re-reads it
//                     from LDS at a shifted offset, so each input element crosses HBM/L2 once per
//                     workgroup instead of once per tap (zero padding, the fused leaky-ReLU /
//                     activation-derivative transforms are applied here, once per element).
//   Each wave owns TM x TN tiles of v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): lane
//   half h feeds channel 2*c2+h, so one instruction consumes two channels of one tap.
//   Epilogue (bias, activation, activation-derivative, residual, mask, accumulate) is fused on
//   the accumulator registers; rows of 32 consecutive positions are written per register.
#include "common.h"
#include "prof.h"

namespace {

struct TileGeom {
  int BKC;   // reduction channels per stage (even)
  int JMAX;  // max taps per phase
  int ROWP;  // LDS pitch of one staged input channel (floats)
  int ntu;   // position tiles per batch element
  int nmt;   // M tiles per group
};

template <int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN)
conv_gemm_kernel(const VcvConvArgs p, const TileGeom tg) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN;
  constexpr int BMP = BM + 1;
  extern __shared__ float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int b = blockIdx.x / tg.ntu, ut = blockIdx.x % tg.ntu;
  const int g = blockIdx.y / tg.nmt, mt = blockIdx.y % tg.nmt;
  const int r = blockIdx.z;

  int J, kw0, kws, oo;
  if (p.phases > 1) {
    kw0 = r; kws = p.phases; oo = p.oo + r;
    J = (r < p.K) ? (p.K - r + p.phases - 1) / p.phases : 0;
  } else {
    kw0 = 0; kws = 1; oo = p.oo; J = p.K;
  }
  const int P = p.P, U = p.Q * P;
  const int u0 = ut * BN, m0 = mt * BM;
  const int qa = u0 / P;
  int qb = (u0 + BN - 1) / P;
  if (qb > p.Q - 1) qb = p.Q - 1;
  const int jspan = (J > 0 ? J - 1 : 0) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0, jmax = jspan > 0 ? jspan : 0;
  const int rlo = qa * p.s + p.off + jmin;
  const int rowlen = ((qb - qa) * p.s + (jmax - jmin) + 1) * P;
  const int BKC = tg.BKC, ROWP = tg.ROWP;
  const int Cg = p.Cg, Mg = p.Mg, K = p.K;

  float* As = smem;
  float* Xs = smem + BKC * tg.JMAX * BMP;

  // per-lane offsets of the B (input) fragment inside one staged channel row
  int laneoff[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int u = u0 + (wn * TN + tn) * 32 + l31;
    if (u > U - 1) u = U - 1;
    const int q = u / P, pc = u - q * P;
    laneoff[tn] = ((q - qa) * p.s - jmin) * P + pc + h * ROWP;
  }

  // a_mode 0: per-lane decode of the (channel, tap) columns of one weight row chunk
  constexpr int NCOL = 4;
  int acol_lds[NCOL], acol_c[NCOL];
  const int ncols = BKC * K;
  if (p.a_mode == 0) {
#pragma unroll
    for (int i = 0; i < NCOL; ++i) {
      const int col = lane + 64 * i;
      acol_lds[i] = -1; acol_c[i] = 0;
      if (col < ncols) {
        const int cl = col / K, kw = col - cl * K;
        const int d = kw - kw0;
        if (d >= 0 && d % kws == 0 && d / kws < J) acol_lds[i] = (cl * J + d / kws) * BMP;
        acol_c[i] = cl;
      }
    }
  }
  const unsigned jmagic = J > 0 ? (1u << 20) / (unsigned)J + 1u : 0u;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  const float* __restrict__ w = p.w;
  const float* __restrict__ x = p.x;
  const long long TinP = (long long)p.Tin * P;
  const long long f0 = (long long)rlo * P;

  for (int c0 = 0; c0 < Cg && J > 0; c0 += BKC) {
    __syncthreads();
    // ---- stage weights ----
    if (p.a_mode == 0) {
      for (int row = wave; row < BM; row += NW) {
        const int m = m0 + row;
        const float* wr = w + ((size_t)(g * Mg + m) * Cg + c0) * K;
#pragma unroll
        for (int i = 0; i < NCOL; ++i) {
          if (acol_lds[i] >= 0) {
            float v = 0.f;
            if (m < Mg && c0 + acol_c[i] < Cg) v = wr[lane + 64 * i];
            As[acol_lds[i] + row] = v;
          }
        }
      }
    } else {
      const int ne = BM * J;
      for (int cl = wave; cl < BKC; cl += NW) {
        const int c = c0 + cl;
        const float* wc = w + ((size_t)(g * Cg + c) * Mg + m0) * K;
        for (int e = lane; e < ne; e += 64) {
          const int ml = (int)(((unsigned)e * jmagic) >> 20);
          const int j = e - ml * J;
          float v = 0.f;
          if (c < Cg && m0 + ml < Mg) v = wc[ml * K + kw0 + j * kws];
          As[(cl * J + j) * BMP + ml] = v;
        }
      }
    }
    // ---- stage input span ----
    for (int cl = wave; cl < BKC; cl += NW) {
      const int c = c0 + cl;
      const size_t base = ((size_t)b * p.G * Cg + (size_t)g * Cg + c) * (size_t)TinP;
      float* xs = Xs + cl * ROWP;
      for (int i = lane; i < rowlen; i += 64) {
        const long long f = f0 + i;
        float v = 0.f;
        if (c < Cg && f >= 0 && f < TinP) {
          v = x[base + f];
          v = vcv_tf(v, p.in_tf, p.xaux, base + f, p.slope);
        }
        xs[i] = v;
      }
    }
    __syncthreads();
    // ---- MFMA over (channel pair, tap) ----
    for (int c2 = 0; c2 < BKC; c2 += 2) {
      const float* Ab = As + (c2 + h) * J * BMP + wm * TM * 32 + l31;
      const float* Xb = Xs + c2 * ROWP;
      for (int j = 0; j < J; ++j) {
        float a[TM], bb[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = Ab[j * BMP + tm * 32];
        const int xo = j * p.dj * P;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bb[tn] = Xb[laneoff[tn] + xo];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      }
    }
  }

  // ---- epilogue ----
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int u = u0 + (wn * TN + tn) * 32 + l31;
    if (u >= U) continue;
    const int q = u / P, pc = u - q * P;
    const int trow = q * p.os + oo;
    if (trow < 0 || trow >= p.Tout) continue;
    const float mk = p.mask ? p.mask[(size_t)b * p.Tout + trow] : 1.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = m0 + (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (ml >= Mg) continue;
        const int mg = g * Mg + ml;
        const size_t idx = (((size_t)b * p.G * Mg + mg) * p.Tout + trow) * P + pc;
        float v = p.alpha * acc[tm][tn][e];
        if (p.bias) v += p.bias[mg];
        v = vcv_act(v, p.out_act, p.slope);
        if (p.out_tf == VCV_TF_DLEAKY) v *= vcv_dleaky(p.oaux[idx], p.slope);
        else if (p.out_tf == VCV_TF_DRELU) v = p.oaux[idx] > 0.f ? v : 0.f;
        else if (p.out_tf == VCV_TF_DTANH) v *= 1.f - p.oaux[idx] * p.oaux[idx];
        if (p.res) v += p.res[idx];
        v *= mk;
        if (p.accumulate) v += p.y[idx];
        p.y[idx] = v;
      }
    }
  }
}

template <int TM, int TN, int WM, int WN>
int launch_conv(const VcvConvArgs& a, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  TileGeom tg;
  const int phases = a.phases > 1 ? a.phases : 1;
  tg.JMAX = phases > 1 ? vcv_cdiv(a.K, phases) : a.K;
  // reduction channels per stage: ~32 (channel, tap) rows, even, at least 2
  int bkc = 32 / tg.JMAX;
  bkc &= ~1;
  if (bkc < 2) bkc = 2;
  if (bkc > 16) bkc = 16;
  while (bkc > 2 && bkc * a.K > 256) bkc -= 2;
  if (a.a_mode == 0 && bkc * a.K > 256) return VCV_EINVAL;
  if (a.a_mode == 1 && BM * tg.JMAX > 4096) return VCV_EINVAL;
  int cg_even = (a.Cg + 1) & ~1;
  if (bkc > cg_even) bkc = cg_even;
  tg.BKC = bkc;
  const int qspan = (BN - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  tg.ROWP = (qspan * a.s + (tg.JMAX - 1) * adj + 1) * a.P;
  const int U = a.Q * a.P;
  tg.ntu = vcv_cdiv(U, BN);
  tg.nmt = vcv_cdiv(a.Mg, BM);
  const size_t lds = ((size_t)tg.BKC * tg.JMAX * (BM + 1) + (size_t)tg.BKC * tg.ROWP) * sizeof(float);
  if (lds > VCV_LDS_LIMIT) return VCV_ELDS;
  auto kern = conv_gemm_kernel<TM, TN, WM, WN>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return VCV_EHIP;
  }
  dim3 grid(a.B * tg.ntu, a.G * tg.nmt, phases), block(64 * WM * WN);
  const double flops = 2.0 * a.B * a.G * a.Mg * a.Cg * a.K * a.P * (double)(phases > 1 ? a.Tin : a.Q);
  const int slot = vcv_prof_start(VCV_PROF_CONV, flops, st);
  hipLaunchKernelGGL(kern, grid, block, lds, st, a, tg);
  vcv_prof_stop(slot, st);
  return vcv_check_launch();
}

}  // namespace

extern "C" int vcv_conv_gemm(const VcvConvArgs* args, void* stream) {
  if (!args) return VCV_EINVAL;
  const VcvConvArgs& a = *args;
  if (a.B <= 0 || a.G <= 0 || a.Cg <= 0 || a.Mg <= 0 || a.Tin <= 0 || a.Tout <= 0 || a.P <= 0 ||
      a.K <= 0 || a.Q <= 0 || a.s <= 0)
    return VCV_EINVAL;
  if (a.in_tf >= VCV_TF_DLEAKY && !a.xaux) return VCV_EINVAL;
  if (a.out_tf != VCV_TF_NONE && !a.oaux) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int U = a.Q * a.P;
  const int phases = a.phases > 1 ? a.phases : 1;
  // tile choice: widest tile that still leaves >= ~2 workgroups per CU and is not mostly padding
  auto ok = [&](int bm, int bn) {
    const int u32 = vcv_cdiv(U, 32) * 32;
    if (bn >= 2 * u32) return false;
    return (long long)a.B * vcv_cdiv(U, bn) * a.G * vcv_cdiv(a.Mg, bm) * phases >= 512;
  };
  if (a.Mg > 64) {
    if (ok(128, 128)) return launch_conv<2, 2, 2, 2>(a, st);
    if (ok(128, 64)) return launch_conv<2, 1, 2, 2>(a, st);
    return launch_conv<1, 1, 2, 2>(a, st);
  }
  if (a.Mg > 32) {
    if (ok(64, 128)) return launch_conv<1, 2, 2, 2>(a, st);
    return launch_conv<1, 1, 2, 2>(a, st);
  }
  if (ok(32, 256)) return launch_conv<1, 2, 1, 4>(a, st);
  return launch_conv<1, 1, 1, 4>(a, st);
}

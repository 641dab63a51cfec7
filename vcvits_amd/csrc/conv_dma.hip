// conv_dma.hip -- LDS-DMA variant of the implicit-GEMM convolution (forward-type launches:
// a_mode 0, one group, no output phases; stride / dilation / period columns as in conv_gemm.hip).
//
// Same GEMM mapping and epilogue as conv_gemm_kernel, different staging: nothing passes through
// registers.
//   * weights are pre-packed per launch into the exact LDS image order
//       wp[m-tile][channel chunk][(c, j)][m]      (zero-padded tails)
//     so one chunk of one tile is a contiguous slab that `global_load_lds_dwordx4` copies 1 KiB per
//     wave-instruction straight into LDS;
//   * the input spans are contiguous per channel and are copied by `buffer_load_dword ... lds`
//     (64 floats per wave-instruction); the descriptor's range check writes zeros for the
//     convolution's padding and for positions past the sequence end;
//   * two LDS buffers: the DMA of chunk c+1 is issued right after the barrier that publishes chunk c
//     and runs under the MFMA loop of chunk c -- one barrier per chunk, no staging VGPRs, no
//     ds_write instructions;
//   * the leaky-ReLU of the input (ResBlocks / generator stages) is applied to the B fragment as
//     it is read from LDS (two VALU ops per fragment element).
#include "common.h"
#include "prof.h"

namespace {

constexpr int ENOFIT = -100;

struct DmaGeom {
  int BKC;      // reduction channels per chunk (even)
  int KKR;      // (channel, tap) rows per chunk = BKC * K
  int nch;      // chunks
  int ntu;      // position tiles per batch element
  int nmt;      // M tiles
  int xw;       // staged span pitch per channel (floats, a multiple of 64)
  int a_floats; // KKR * BM
  int buf_floats;  // a_floats + BKC * XW
  int JA;          // taps per channel stored in a weight slab (K, or ceil(K/phases) for a phased launch)
  int phases;      // > 1: transposed / strided-data-gradient launch, one residue per blockIdx.z
  int ks;          // > 1: the chunks are split over ks blocks per tile, partial sums go to a scratch slab each
};

typedef __attribute__((address_space(3))) void* lds_ptr;

// ---- weight pack: w [M, C, K] -> wp [nmt][nch][KKR][BM] ------------------------------------------------
// flip != 0 packs the data-gradient view instead: A(m, c, k) = w[c, m, K-1-k]  (w is then [C, M, K])
template <int BM>
__global__ void __launch_bounds__(256)
pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int M, int C, int K, int BKC, int KKR,
                    int nch, int flip, int rsplit) {
  extern __shared__ float t[];  // [KKR][BM + 1]
  // blockIdx.y = m-tile * rsplit + row slice: small weights are split over more workgroups (RB rows each)
  const int ch = blockIdx.x, mt = blockIdx.y / rsplit, rs = blockIdx.y - mt * rsplit;
  const int RB = BM / rsplit, r0 = rs * RB;
  const int nmt = gridDim.y / rsplit;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = ch * BKC, m0 = mt * BM;
  if (!flip) {
#pragma unroll 8
    for (int ml = r0 + wave; ml < r0 + RB; ml += 4) {
      const int m = m0 + ml;
      const float* wr = w + ((size_t)m * C + c0) * K;
      for (int kk = lane; kk < KKR; kk += 64) {
        const int c = c0 + kk / K;
        t[kk * (BM + 1) + ml] = (m < M && c < C) ? wr[kk] : 0.f;
      }
    }
  } else if (flip == 1) {
    // w[c, m, k]: for fixed c the (m, k) block is contiguous
    for (int cl = wave; cl < BKC; cl += 4) {
      const int c = c0 + cl;
      const float* wc = w + ((size_t)c * M + m0) * K;
#pragma unroll 8
      for (int e = r0 * K + lane; e < (r0 + RB) * K; e += 64) {
        const int ml = e / K, k = e - ml * K;
        t[(cl * K + (K - 1 - k)) * (BM + 1) + ml] = (c < C && m0 + ml < M) ? wc[e] : 0.f;
      }
    }
  } else {
    // phased (transposed) launch: residue r = blockIdx.z keeps taps k = r + j*phases of w[c, m, k]; here the
    // argument K is the weight's tap count and KKR = BKC * JA
    const int phases = flip >> 8, r = blockIdx.z, JA = KKR / BKC;
    for (int i = tid; i < KKR * (BM + 1); i += 256) t[i] = 0.f;
    __syncthreads();
    for (int cl = wave; cl < BKC; cl += 4) {
      const int c = c0 + cl;
      const float* wc = w + ((size_t)c * M + m0) * K;
      for (int e = r0 * K + lane; e < (r0 + RB) * K; e += 64) {
        const int ml = e / K, k = e - ml * K;
        if (k % phases == r && c < C && m0 + ml < M) t[(cl * JA + k / phases) * (BM + 1) + ml] = wc[e];
      }
    }
  }
  __syncthreads();
  float* out = wp + (((size_t)blockIdx.z * nmt + mt) * nch + ch) * (size_t)KKR * BM;
  // RB is a multiple of 4 and the slab is 16-byte aligned: 16-byte stores
  for (int i = tid * 4; i < KKR * RB; i += 1024) {
    const int kk = i / RB, ml = r0 + (i - kk * RB);
    const float* tr = t + kk * (BM + 1) + ml;
    *reinterpret_cast<float4*>(out + kk * BM + ml) = make_float4(tr[0], tr[1], tr[2], tr[3]);
  }
}

template <int TM, int TN, int WM, int WN, bool LEAKY>
__global__ void __launch_bounds__(64 * WM * WN)
conv_dma_kernel(const VcvConvArgs p, const DmaGeom tg, const float* __restrict__ wp, float* __restrict__ part) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN;
  extern __shared__ float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int kz = blockIdx.x % tg.ks;
  const int bx = blockIdx.x / tg.ks;
  const int b = bx / tg.ntu, ut = bx % tg.ntu;
  const int mt = blockIdx.y;
  const int r = blockIdx.z;  // output residue of a phased launch (0 otherwise)
  const int JA = tg.JA, P = p.P, U = p.Q * P, Cg = p.Cg, Mg = p.Mg;
  // taps of this residue: k = r + j*phases < K
  const int K = tg.phases > 1 ? (r < p.K ? (p.K - r + tg.phases - 1) / tg.phases : 0) : p.K;
  const int oo = p.oo + (tg.phases > 1 ? r : 0);
  const int u0 = ut * BN, m0 = mt * BM;
  const int qa = u0 / P;
  const int jspan = (JA - 1) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0;
  const int f0 = (qa * p.s + p.off + jmin) * P;
  const int BKC = tg.BKC, KKR = tg.KKR;
  const int XW = tg.xw;

  int laneoff[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int u = u0 + (wn * TN + tn) * 32 + l31;
    if (u > U - 1) u = U - 1;
    const int q = u / P, pc = u - q * P;
    laneoff[tn] = ((q - qa) * p.s - jmin) * P + pc + h * XW;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  const long long TinP = (long long)p.Tin * P;
  const float* xb = p.x + (size_t)b * Cg * (size_t)TinP;
  const float* wtile = wp + ((size_t)r * gridDim.y + mt) * tg.nch * (size_t)tg.a_floats;
  const int nA = tg.a_floats >> 8;       // 1 KiB wave-instructions per weight slab
  const int nXrow = XW >> 6;             // 256-B wave-instructions per channel span
  const int nX = BKC * nXrow;

  auto issue = [&](int ch, int buf) {
    float* As = smem + buf * tg.buf_floats;
    float* Xs = As + tg.a_floats;
    const float* slab = wtile + (size_t)ch * tg.a_floats;
    for (int i = wave; i < nA; i += NW)
      __builtin_amdgcn_global_load_lds(slab + i * 256 + lane * 4, (lds_ptr)(As + i * 256), 16, 0, 0);
    const int c0 = ch * BKC;
    for (int i = wave; i < nX; i += NW) {
      const int cl = i / nXrow, part = i - cl * nXrow;
      const int c = c0 + cl;
      const unsigned rec = c < Cg ? (unsigned)(TinP * 4) : 0u;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)c * (size_t)TinP), 0, (int)rec, 0x00020000);
      const unsigned voff = (unsigned)(f0 + part * 64 + lane) * 4u;  // negative -> wraps -> out of range -> 0
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(Xs + cl * XW + part * 64), 4, voff, 0, 0, 0);
    }
  };

  const int ch_begin = (int)((long long)kz * tg.nch / tg.ks), ch_end = (int)((long long)(kz + 1) * tg.nch / tg.ks);
  issue(ch_begin, 0);
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    // publish chunk ch (its DMA is the only traffic in flight here) and retire every wave's reads of the
    // other buffer before it is overwritten
    __syncthreads();
    const int cb = (ch - ch_begin) & 1;
    if (ch + 1 < ch_end) issue(ch + 1, cb ^ 1);
    const float* As = smem + cb * tg.buf_floats;
    const float* Xs = As + tg.a_floats;
    for (int c2 = 0; c2 < BKC; c2 += 2) {
      const float* Ab = As + (c2 + h) * JA * BM + wm * TM * 32 + l31;
      const float* Xb = Xs + c2 * XW;
      for (int j = 0; j < K; ++j) {
        float a[TM], bb[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = Ab[j * BM + tm * 32];
        const int xo = j * p.dj * P;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          float v = Xb[laneoff[tn] + xo];
          if (LEAKY) v = fmaxf(v, v * p.slope);  // slope < 1
          bb[tn] = v;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      }
    }
  }

  const int rows_valid = Mg - m0 < BM ? Mg - m0 : BM;
  const bool mtail = m0 + BM > Mg;
  if (tg.ks > 1) {
    // split reduction: raw partial sums to this split's slab [kz][b][m][u]; conv_dma_finish_kernel adds the
    // slabs and applies the epilogue
    float* pb = part + (((size_t)kz * p.B + b) * Mg + m0) * (size_t)U;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int u = u0 + (wn * TN + tn) * 32 + l31;
      if (u >= U) continue;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ml = (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (mtail && ml >= rows_valid) continue;
          pb[(size_t)ml * U + u] = acc[tm][tn][e];
        }
    }
    return;
  }
  // ---- epilogue (as conv_gemm_kernel) ----
  const unsigned rowstride = (unsigned)(p.Tout * P);
  const size_t ybase = ((size_t)b * Mg + m0) * rowstride;
  const float* bias = p.bias ? p.bias + m0 : nullptr;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int u = u0 + (wn * TN + tn) * 32 + l31;
    if (u >= U) continue;
    const int q = u / P, pc = u - q * P;
    const int trow = q * p.os + oo;
    if (trow < 0 || trow >= p.Tout) continue;
    const float mk = p.mask ? p.mask[(size_t)b * p.Tout + trow] : 1.f;
    const size_t colbase = ybase + (size_t)trow * P + pc;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (mtail && ml >= rows_valid) continue;
        const size_t idx = colbase + (size_t)((unsigned)ml * rowstride);
        float v = p.alpha * acc[tm][tn][e];
        if (bias) v += bias[ml];
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

// Adds the ks partial slabs of a split launch and applies the epilogue of conv_dma_kernel.
__global__ void __launch_bounds__(256) conv_dma_finish_kernel(const VcvConvArgs p, const float* __restrict__ part, int ks) {
  const int U = p.Q * p.P;
  const size_t n = (size_t)p.B * p.Mg * U;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int u = (int)(i % U);
  const size_t bm = i / U;
  const int m = (int)(bm % p.Mg), b = (int)(bm / p.Mg);
  const int q = u / p.P, pc = u - q * p.P;
  const int trow = q * p.os + p.oo;
  if (trow < 0 || trow >= p.Tout) return;
  float v = 0.f;
  for (int k = 0; k < ks; ++k) v += part[(size_t)k * n + i];
  v *= p.alpha;
  if (p.bias) v += p.bias[m];
  v = vcv_act(v, p.out_act, p.slope);
  const size_t idx = (bm * p.Tout + trow) * p.P + pc;
  if (p.out_tf == VCV_TF_DLEAKY) v *= vcv_dleaky(p.oaux[idx], p.slope);
  else if (p.out_tf == VCV_TF_DRELU) v = p.oaux[idx] > 0.f ? v : 0.f;
  else if (p.out_tf == VCV_TF_DTANH) v *= 1.f - p.oaux[idx] * p.oaux[idx];
  if (p.res) v += p.res[idx];
  if (p.mask) v *= p.mask[(size_t)b * p.Tout + trow];
  if (p.accumulate) v += p.y[idx];
  p.y[idx] = v;
}

struct Plan {
  int variant;  // 8: 128x128 with 8 waves (2 per SIMD when only one workgroup fits / exists per CU)
                // 0: 128x128 (4 waves)  1: 128x256 (8 waves)  2: 128x224 (7 waves)  3: 64x224 (7 waves)
                // 4: 64x256 (8 waves: 2x2 per wave, 1x4 waves... see launch)  5: 64x128
  int BM, BN;
  DmaGeom g;
  size_t ws_floats, pack_floats, lds_bytes;
};

bool eligible(const VcvConvArgs& a) {
  const bool fwd_type = a.a_mode == 0 && a.phases <= 1;
  const bool phased = a.a_mode == 1 && a.phases > 1 && a.s == 1 && a.dj == -1;
  return (fwd_type || phased) && a.G == 1 && a.io == 0 && a.post_scale == 0.f && a.ms <= 1 && (a.in_tf == VCV_TF_NONE || (a.in_tf == VCV_TF_LEAKY && a.slope < 1.f && a.slope >= 0.f)) &&
         a.Mg >= 32 && a.Cg >= 16 && a.K <= 16 && a.s >= 1 && (long long)a.Tin * a.P * 4 < (1ll << 31) &&
         (long long)a.Mg * a.Tout * a.P < (1ll << 31);
}

bool make_plan(const VcvConvArgs& a, int BM, int BN, Plan& pl) {
  pl.BM = BM; pl.BN = BN;
  DmaGeom& g = pl.g;
  const int qspan = (BN - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  g.phases = a.phases > 1 ? a.phases : 1;
  g.JA = vcv_cdiv(a.K, g.phases);
  const int rowmax = (qspan * a.s + (g.JA - 1) * adj + 1) * a.P;
  const int xw = (rowmax + 63) & ~63;
  g.xw = xw;
  // chunk: ~64 (c, tap) rows, even channel count, KKR*BM a multiple of 256 floats, two buffers within ~120 KiB
  int bkc = 64 / g.JA;
  bkc &= ~1;
  if (bkc < 2) bkc = 2;
  const int cg_even = (a.Cg + 1) & ~1;
  if (bkc > cg_even) bkc = cg_even;
  const size_t lds_cap = 78 * 1024;  // two blocks per CU
  while (bkc > 2 && 2ull * ((size_t)bkc * g.JA * BM + (size_t)bkc * xw) * 4 > lds_cap) bkc -= 2;
  // weight slabs move as whole 1-KiB DMA instructions: nearest even channel count (down, else up) that fits
  {
    int dn = bkc, up = bkc;
    while (dn >= 2 && (dn * g.JA * BM) % 256 != 0) dn -= 2;
    while (up <= 2 * bkc + 8 && (up * g.JA * BM) % 256 != 0) up += 2;
    if (dn >= 2) bkc = dn;
    else if ((up * g.JA * BM) % 256 == 0) bkc = up;
    else return false;
  }
  g.BKC = bkc;
  g.KKR = bkc * g.JA;
  g.nch = vcv_cdiv(a.Cg, bkc);
  g.ntu = vcv_cdiv(a.Q * a.P, BN);
  g.nmt = vcv_cdiv(a.Mg, BM);
  g.a_floats = g.KKR * BM;
  g.buf_floats = g.a_floats + bkc * xw;
  pl.lds_bytes = 2ull * g.buf_floats * 4;
  if (pl.lds_bytes > VCV_LDS_LIMIT) return false;
  g.ks = 1;
  pl.pack_floats = (size_t)g.phases * g.nmt * g.nch * g.a_floats;
  pl.ws_floats = pl.pack_floats;
  return true;
}

bool choose(const VcvConvArgs& a, Plan& pl) {
  const int U = a.Q * a.P;
  // measured on the bench step: the packed / DMA path wins for deep reductions over long rows (k >= 5, or
  // k >= 3 for <= 64 output channels, and >= 160 positions per batch element); short rows and wide k = 3
  // layers stay on the register-staged kernel (or take the split mode below)
  // k = 1 / 2 (pointwise convs of the WaveNet stacks, flow and attention projections): plain GEMMs, taken when the
  // reduction is deep enough to amortise the staging
  const bool pointwise_ok = a.K <= 2 && a.Cg * a.K >= 128;
  bool normal_ok = a.phases > 1 ? (a.K >= 4 && U >= 160) : ((a.K >= (a.Mg <= 64 ? 3 : 5) || pointwise_ok) && U >= 160);
  if (!normal_ok && (a.phases > 1 || (a.K < 3 && !pointwise_ok) || U < 128)) return false;
  const int nph = a.phases > 1 ? a.phases : 1;
  auto blocks = [&](int bm, int bn) { return (long long)a.B * vcv_cdiv(U, bn) * vcv_cdiv(a.Mg, bm) * nph; };
  if (normal_ok && U > 160 && U <= 224) {
    if (a.Mg >= 128 && blocks(128, 224) >= (nph > 1 ? 128 : 224) && make_plan(a, 128, 224, pl)) { pl.variant = 2; return true; }
    if (a.Mg >= 64 && blocks(64, 224) >= (a.Mg >= 128 ? 192 : 1) && make_plan(a, 64, 224, pl)) { pl.variant = 3; return true; }
  }
  // rows just past 256 positions (period 37 strided gradients: 259): one 288-wide tile, not 256 + 3
  if (normal_ok && U > 256 && U <= 288 && a.Mg >= 128 && blocks(128, 288) >= 128 && make_plan(a, 128, 288, pl)) {
    pl.variant = 9;
    return true;
  }
  if (normal_ok && a.Mg >= 128) {
    if (U > 160 && blocks(128, 256) >= 256 && make_plan(a, 128, 256, pl)) { pl.variant = 1; return true; }
    if (blocks(128, 128) >= (nph > 1 ? 128 : 256) && make_plan(a, 128, 128, pl)) { pl.variant = blocks(128, 128) < 512 ? 8 : 0; return true; }
  }
  if (normal_ok && a.Mg >= 64 && !(a.Mg >= 128 && blocks(128, 128) >= 32)) {
    if (a.K < 5 && U > 160 && blocks(64, 256) >= 256 && make_plan(a, 64, 256, pl)) { pl.variant = 4; return true; }
    if (make_plan(a, 64, 128, pl) && blocks(64, 128) >= 256) { pl.variant = 5; return true; }
  }
  if (normal_ok && a.Mg >= 32 && a.Mg < 64 && blocks(32, 256) >= 256 && make_plan(a, 32, 256, pl)) { pl.variant = 6; return true; }
  // too few tiles to fill the chip: split the reduction over ks blocks per tile (deterministic slabs + a
  // finishing pass), which also hides the DMA latency the few resident waves cannot
  if (nph == 1 && a.Mg >= 64) {
    bool ok = false;
    if (U > 128 && U <= 224) {
      if (a.Mg >= 128 && blocks(128, 224) >= 32 && make_plan(a, 128, 224, pl)) pl.variant = 2, ok = true;
      else if (make_plan(a, 64, 224, pl)) pl.variant = 3, ok = true;
    }
    if (!ok && a.Mg >= 128 && blocks(128, 128) >= 32 && make_plan(a, 128, 128, pl)) pl.variant = 8, ok = true;
    if (!ok && make_plan(a, 64, 128, pl)) pl.variant = 5, ok = true;
    if (!ok) return false;
    const long long nb = blocks(pl.BM, pl.BN);
    long long ks = (384 + nb - 1) / nb;
    if (ks > pl.g.nch / 2) ks = pl.g.nch / 2;
    if (ks < 2) return false;
    pl.g.ks = (int)ks;
    pl.ws_floats = pl.pack_floats + (size_t)ks * a.B * a.Mg * U;
    return true;
  }
  return false;
}

template <int TM, int TN, int WM, int WN>
int launch(const VcvConvArgs& a, const Plan& pl, float* ws, float* part, bool pack_valid, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, NT = 64 * WM * WN;
  const DmaGeom& g = pl.g;
  // pack (forward orientation; the data-gradient orientation is packed by the caller through flip)
  const size_t plds = (size_t)g.KKR * (BM + 1) * 4;
  auto pk = pack_weights_kernel<BM>;
  if (plds > 64 * 1024 && hipFuncSetAttribute((const void*)pk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds) != hipSuccess)
    return VCV_EHIP;
  const int flip = g.phases > 1 ? (g.phases << 8) : (a.accumulate >> 8);
  if (!pack_valid) {
    // small weights: split the rows of a tile over up to BM/32 workgroups so the pack is not a 16-workgroup launch
    int rsplit = 1;
    while (rsplit * 32 < BM && (long long)g.nch * g.nmt * g.phases * rsplit < 256) rsplit *= 2;
    hipLaunchKernelGGL(pk, dim3(g.nch, g.nmt * rsplit, g.phases), dim3(256), plds, st, a.w, ws, a.Mg, a.Cg, a.K, g.BKC,
                       g.KKR, g.nch, flip, rsplit);
  }
  void (*kern)(const VcvConvArgs, const DmaGeom, const float*, float*) =
      a.in_tf == VCV_TF_LEAKY ? conv_dma_kernel<TM, TN, WM, WN, true> : conv_dma_kernel<TM, TN, WM, WN, false>;
  if (pl.lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes) != hipSuccess)
    return VCV_EHIP;
  VcvConvArgs aa = a;
  aa.accumulate = a.accumulate & 1;
  dim3 grid(a.B * g.ntu * g.ks, g.nmt, g.phases), block(NT);
  const double flops = 2.0 * a.B * a.Mg * a.Cg * a.K * a.P * (double)(g.phases > 1 ? a.Tin : a.Q);
  const int tag[12] = {a.B, 1, a.Cg, a.Mg, a.K, a.Q, a.P, a.s, g.phases, a.a_mode + 10 * g.ks, BM * 1000 + pl.BN, g.BKC};
  hipEvent_t ev0, ev1;
  // algorithmic bytes: input once, weights once, output once (+ the fused epilogue operands)
  const double abytes = 4.0 * ((double)a.B * a.Cg * a.Tin * a.P + (double)a.Mg * a.Cg * a.K +
                               (double)a.B * a.Mg * a.Tout * a.P * (1 + (a.res ? 1 : 0) + (a.oaux ? 1 : 0)));
  vcv_prof_events(VCV_PROF_CONV_DMA, flops, tag, 12, &ev0, &ev1, abytes);
  VCV_LAUNCH_EV(kern, grid, block, (unsigned)pl.lds_bytes, st, ev0, ev1, aa, g, (const float*)ws, part);
  if (g.ks > 1) {
    const size_t n = (size_t)a.B * a.Mg * a.Q * a.P;
    hipLaunchKernelGGL(conv_dma_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, aa, (const float*)part, g.ks);
  }
  return vcv_check_launch();
}

}  // namespace

// Workspace (floats) the DMA path needs for this launch, 0 if the launch is not eligible.
extern "C" int64_t vcv_conv_dma_workspace(const VcvConvArgs* args) {
  if (!args || !eligible(*args)) return 0;
  Plan pl;
  if (!choose(*args, pl)) return 0;
  return (int64_t)pl.ws_floats;
}

namespace {
int run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, bool pack_valid, void* stream) {
  if (!args || !pack_ws || !eligible(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose(*args, pl)) return VCV_EINVAL;
  if (pl.g.ks > 1 && !scratch_ws) return VCV_EINVAL;
  VcvConvArgs a = *args;
  a.accumulate = (a.accumulate & 1) | (flip ? 256 : 0);
  hipStream_t st = (hipStream_t)stream;
  // more, smaller waves where measured faster: 14 waves on the 128x224 tile (non-phased), 8 on 32x256
  if (pl.variant == 2 && pl.g.phases == 1) pl.variant = 12;
  if (pl.variant == 6) pl.variant = 16;
  if (pl.variant == 1) pl.variant = 21;  // 128x256 with 16 waves
  if (pl.variant == 3) pl.variant = 23;  // 64x224 with 14 waves
  if (pl.variant == 5) pl.variant = 25;  // 64x128 with 8 waves
  switch (pl.variant) {
    case 0: return launch<2, 2, 2, 2>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 2: return launch<4, 1, 1, 7>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 4: return launch<2, 2, 1, 4>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 8: return launch<2, 1, 2, 4>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 9: return launch<4, 1, 1, 9>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 12: return launch<2, 1, 2, 7>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 16: return launch<1, 1, 1, 8>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 21: return launch<2, 1, 2, 8>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    case 23: return launch<1, 1, 2, 7>(a, pl, pack_ws, scratch_ws, pack_valid, st);
    default: return launch<1, 1, 2, 4>(a, pl, pack_ws, scratch_ws, pack_valid, st);
  }
}
}  // namespace

// `flip`: 0 = w is [M, C, K] (forward); 1 = w is [C, M, K] and the launch is the stride-1 data gradient
// (the pack applies the flip / transpose, so no separate vcv_weight_flip_transpose pass is needed).
extern "C" int vcv_conv_dma(const VcvConvArgs* args, float* workspace, int flip, void* stream) {
  if (!args || !workspace || !eligible(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose(*args, pl)) return VCV_EINVAL;
  return run(args, workspace, workspace + pl.pack_floats, flip, false, stream);
}

// Split form for callers that keep packed weights across launches.  vcv_conv_dma_plan: out[0] = floats of the
// packed-weight buffer, out[1] = floats of per-launch scratch (0 unless the reduction is split), out[2] = a
// signature of the pack layout (tile height, chunk size, taps, phases, flip): a packed buffer may be reused
// by any later launch over the SAME unchanged weights whose plan has the same signature and pack size.
// Returns 0, or VCV_EINVAL when the launch is not eligible.  vcv_conv_dma_run: pack_valid != 0 skips the pack.
extern "C" int vcv_conv_dma_plan(const VcvConvArgs* args, int flip, int64_t* out) {
  if (!args || !out || !eligible(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose(*args, pl)) return VCV_EINVAL;
  out[0] = (int64_t)pl.pack_floats;
  out[1] = (int64_t)(pl.ws_floats - pl.pack_floats);
  const DmaGeom& g = pl.g;
  out[2] = ((int64_t)pl.BM << 40) | ((int64_t)g.BKC << 28) | ((int64_t)g.JA << 20) | ((int64_t)g.phases << 8) | (flip ? 1 : 0);
  return 0;
}

extern "C" int vcv_conv_dma_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                                void* stream) {
  return run(args, pack_ws, scratch_ws, flip, pack_valid != 0, stream);
}

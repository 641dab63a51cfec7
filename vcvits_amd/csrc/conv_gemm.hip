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
//   Staging is software-pipelined through registers: the global loads of chunk c+1 are issued
//   before the MFMA loop of chunk c and written to LDS after it, so HBM/L2 latency hides behind
//   the matrix pipe.  All index decodes use power-of-two widths (shifts), fixed per thread.
//   Each wave owns TM x TN tiles of v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD): lane
//   half h feeds channel 2*c2+h, so one instruction consumes two channels of one tap.
//   Epilogue (bias, activation, activation-derivative, residual, mask, accumulate) is fused on
//   the accumulator registers; rows of 32 consecutive positions are written per register.
#include "common.h"
#include "prof.h"

namespace {

constexpr int VCV_ENOFIT = -100;  // internal: tile geometry exceeds the prefetch budget
constexpr int APT_DEFAULT = 20;  // max weight elements prefetched per thread per chunk
constexpr int APT_SMALL_TILE = 32;  // ... for single-accumulator-tile waves (room in the register file)
constexpr int XPT_DEFAULT = 12;  // max input elements prefetched per thread per chunk
constexpr int XPT_WIDE = 24;     // ... for the 7-wave tiles (small accumulator footprint, wide strided spans)

// input-transform specialisations (template parameter INTF)
constexpr int INTF_NONE = 0, INTF_LEAKY = 1, INTF_DLEAKY = 2, INTF_AUX = 3;

struct TileGeom {
  int BKC;       // reduction channels per stage (even)
  int JMAX;      // max taps per phase
  int ntu;       // position tiles per batch element
  int nmt;       // M tiles per group
  int BMP;       // LDS pitch of one weight row (BM + pad)
  int cw_log;    // a_mode 0: log2 of the padded (channel, tap) row width; a_mode 1: log2 of padded taps
  int xw_log;    // log2 of the staged-span pitch (>= 6 so one wave-instruction stays inside one channel)
  int napass;    // weight passes per thread (<= APT)
  int nxpass;    // input passes per thread (<= XPT)
  int a_floats;  // LDS floats reserved for the weight tile (incl. never-read overshoot rows)
  int xsync;     // 1: spans too wide for the prefetch registers -> staged synchronously (rare: stride >= 8)
};

__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

// Every global read of the staging path is a buffer load: the descriptor's range check returns 0
// for rows / channels / time positions outside the tensor, so zero padding, ragged tiles and the
// channel tail cost no predicate instructions.
template <int TM, int TN, int WM, int WN, int INTF>
__global__ void __launch_bounds__(64 * WM * WN, 2)
conv_gemm_kernel(const VcvConvArgs p, const TileGeom tg) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN, NT = 64 * NW;
  constexpr int BM_LOG = (BM == 128) ? 7 : (BM == 64 ? 6 : 5);
  constexpr bool XAUX = INTF >= INTF_DLEAKY;
  constexpr int APT = (TM * TN == 1) ? APT_SMALL_TILE : APT_DEFAULT;
  constexpr int XPT = (WN == 7 || TN * WN >= 8) ? (XAUX ? 16 : XPT_WIDE) : XPT_DEFAULT;
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
  const int jspan = (J > 0 ? J - 1 : 0) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0;
  const int rlo = qa * p.s + p.off + jmin;
  const int BKC = tg.BKC, BMP = tg.BMP;
  const int Cg = p.Cg, Mg = p.Mg, K = p.K;
  const int XW = 1 << tg.xw_log;

  float* As = smem;
  float* Xs = smem + tg.a_floats;

  // per-lane offsets of the B (input) fragment inside one staged channel row
  int laneoff[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int u = u0 + (wn * TN + tn) * 32 + l31;
    if (u > U - 1) u = U - 1;
    const int q = u / P, pc = u - q * P;
    laneoff[tn] = ((q - qa) * p.s - jmin) * P + pc + h * XW;
  }

  // ---- per-thread staging maps (chunk-invariant, power-of-two widths) ----
  const int cw = 1 << tg.cw_log;
  const int a_col = tid & (cw - 1);
  const int a_row0 = tid >> tg.cw_log;
  const int a_rstep = NT >> tg.cw_log;
  const unsigned rowpitch = (unsigned)(Cg * K);
  int a_lds = -1, a_cl = 0, a_kw = 0;
  unsigned a_voff = 0xFFFFFFFFu;
  if (p.a_mode == 0) {
    // thread owns column a_col of the (channel, tap) row chunk, rows a_row0 + i*a_rstep
    if (a_col < BKC * K) {
      a_cl = a_col / K;
      a_kw = a_col - a_cl * K;
      const int d = a_kw - kw0;
      if (d >= 0 && d % kws == 0 && d / kws < J) {
        a_lds = (a_cl * J + d / kws) * BMP + a_row0;
        a_voff = ((unsigned)a_row0 * rowpitch + (unsigned)a_col) * 4u;
      }
    }
  } else {
    // thread owns tap slot a_col, flat (channel, m) pairs a_row0 + i*a_rstep
    if (a_col < J) a_lds = a_col * BMP;
    a_kw = kw0 + a_col * kws;
  }
  const unsigned a_vstep = (unsigned)a_rstep * rowpitch * 4u;
  const bool mtail = m0 + BM > Mg;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  const long long TinP = (long long)p.Tin * P;
  const unsigned xrec = (unsigned)(TinP * 4);
  const size_t xbase_b = ((size_t)b * p.G * Cg + (size_t)g * Cg) * (size_t)TinP;
  const int f0 = rlo * P;
  const int rows_valid = Mg - m0 < BM ? Mg - m0 : BM;

  float areg[APT];
  float xreg[XPT];
  float xareg[XAUX ? XPT : 1];
  int cur_c0 = 0;

  auto load_chunk = [&](int c0) {
    if (p.a_mode == 0) {
      const float* base = p.w + (size_t)(g * Mg + m0) * rowpitch + (size_t)c0 * K;
      const long long rem = (long long)rows_valid * rowpitch - (long long)c0 * K;
      __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(rem > 0 ? rem * 4 : 0), 0x00020000);
      const unsigned v0 = (c0 + a_cl < Cg) ? a_voff : 0xFFFFFFFFu;
#pragma unroll
      for (int i = 0; i < APT; ++i)
        if (i < tg.napass) areg[i] = ld_buf(ra, v0 == 0xFFFFFFFFu ? v0 : v0 + (unsigned)i * a_vstep);
    } else {
      const float* base = p.w + ((size_t)(g * Cg + c0) * Mg + m0) * K;
      const long long rem = ((long long)(Cg - c0) * Mg - m0) * K;
      __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(rem > 0 ? rem * 4 : 0), 0x00020000);
#pragma unroll
      for (int i = 0; i < APT; ++i)
        if (i < tg.napass) {
          const int f = a_row0 + i * a_rstep;  // flat (cl, m)
          const int cl = f >> BM_LOG, ml = f & (BM - 1);
          unsigned v = ((unsigned)(cl * Mg + ml) * (unsigned)K + (unsigned)a_kw) * 4u;
          if (a_lds < 0 || (mtail && ml >= rows_valid)) v = 0xFFFFFFFFu;
          areg[i] = ld_buf(ra, v);
        }
    }
    cur_c0 = c0;
    if (!tg.xsync)
#pragma unroll
    for (int i = 0; i < XPT; ++i)
      if (i < tg.nxpass) {
        const int f = tid + i * NT;
        const int cl = __builtin_amdgcn_readfirstlane(f >> tg.xw_log);
        const int col = f & (XW - 1);
        const size_t rowbase = xbase_b + (size_t)(c0 + cl) * (size_t)TinP;
        const unsigned rec = (c0 + cl < Cg) ? xrec : 0u;
        const unsigned voff = (unsigned)(f0 + col) * 4u;  // negative positions wrap -> out of range -> 0
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + rowbase), 0, (int)rec, 0x00020000);
        xreg[i] = ld_buf(rx, voff);
        if (XAUX) {
          __amdgpu_buffer_rsrc_t rxa = __builtin_amdgcn_make_buffer_rsrc((void*)(p.xaux + rowbase), 0, (int)rec, 0x00020000);
          xareg[i] = ld_buf(rxa, voff);
        }
      }
  };

  auto store_chunk = [&]() {
    if (a_lds >= 0) {
      if (p.a_mode == 0) {
#pragma unroll
        for (int i = 0; i < APT; ++i)
          if (i < tg.napass) As[a_lds + i * a_rstep] = areg[i];
      } else {
#pragma unroll
        for (int i = 0; i < APT; ++i)
          if (i < tg.napass) {
            const int f = a_row0 + i * a_rstep;
            const int cl = f >> BM_LOG, ml = f & (BM - 1);
            As[cl * J * BMP + a_lds + ml] = areg[i];
          }
      }
    }
    if (tg.xsync) {
      const int nel = BKC << tg.xw_log;
      for (int f = tid; f < nel; f += NT) {
        const int cl = f >> tg.xw_log, col = f & (XW - 1);
        const long long ff = (long long)f0 + col;
        float v = 0.f;
        if (cur_c0 + cl < Cg && ff >= 0 && ff < TinP) {
          const size_t gi = xbase_b + (size_t)(cur_c0 + cl) * (size_t)TinP + (size_t)ff;
          v = vcv_tf(p.x[gi], p.in_tf, p.xaux, gi, p.slope);
        }
        Xs[f] = v;
      }
    } else
#pragma unroll
    for (int i = 0; i < XPT; ++i)
      if (i < tg.nxpass) {
        float v = xreg[i];
        if (INTF == INTF_LEAKY) v = vcv_leaky(v, p.slope);
        if (INTF == INTF_DLEAKY) v *= vcv_dleaky(xareg[i], p.slope);
        if (INTF == INTF_AUX) {
          const float av = xareg[i];
          if (p.in_tf == VCV_TF_DRELU) v = av > 0.f ? v : 0.f;
          else if (p.in_tf == VCV_TF_DTANH) v *= 1.f - av * av;
          else if (p.in_tf == VCV_TF_DLOGCLAMP) v = av > logf(p.slope) ? v * expf(-av) : 0.f;
          else v *= vcv_dleaky(av, p.slope);
        }
        Xs[tid + i * NT] = v;  // == Xs[cl * XW + col]
      }
  };

  if (J > 0) {
    load_chunk(0);
    store_chunk();
    __syncthreads();
    for (int c0 = 0; c0 < Cg; c0 += BKC) {
      const bool more = c0 + BKC < Cg;
      if (more) load_chunk(c0 + BKC);
      // ---- MFMA over (channel pair, tap) ----
      for (int c2 = 0; c2 < BKC; c2 += 2) {
        const float* Ab = As + (c2 + h) * J * BMP + wm * TM * 32 + l31;
        const float* Xb = Xs + c2 * XW;
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
      if (more) {
        __syncthreads();
        store_chunk();
        __syncthreads();
      }
    }
  }

  // ---- epilogue ----
  const unsigned rowstride = (unsigned)(p.Tout * P);
  const size_t ybase = ((size_t)b * p.G * Mg + (size_t)g * Mg + m0) * rowstride;
  const float* bias = p.bias ? p.bias + g * Mg + m0 : nullptr;
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

inline int ilog2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

template <int TM, int TN, int WM, int WN>
int launch_conv(const VcvConvArgs& a, hipStream_t st, bool allow_sync = false) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
  constexpr int APT = (TM * TN == 1) ? APT_SMALL_TILE : APT_DEFAULT;
  const int XPT = (WN == 7 || TN * WN >= 8) ? (a.in_tf >= VCV_TF_DLEAKY ? 16 : XPT_WIDE) : XPT_DEFAULT;
  TileGeom tg;
  const int phases = a.phases > 1 ? a.phases : 1;
  tg.JMAX = phases > 1 ? vcv_cdiv(a.K, phases) : a.K;
  const int qspan = (BN - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  const int rowmax = (qspan * a.s + (tg.JMAX - 1) * adj + 1) * a.P;
  tg.xw_log = ilog2_ceil(rowmax);
  if (tg.xw_log < 6) tg.xw_log = 6;
  const int xw = 1 << tg.xw_log;
  const int cg_even = (a.Cg + 1) & ~1;
  const int min_cwl = ilog2_ceil(vcv_cdiv(NT, BM));  // at least NT/BM columns so one pass never overshoots BM rows
  // largest even channel chunk whose prefetch fits the per-thread register budget, <= ~64 (c,tap) rows
  int best = 0;
  for (int bkc = 2; bkc <= 64 && bkc <= cg_even; bkc += 2) {
    if (bkc * tg.JMAX > 64 && bkc > 2) break;
    int napass;
    if (a.a_mode == 0) {
      int cwl = ilog2_ceil(bkc * a.K);
      if (cwl < min_cwl) cwl = min_cwl;
      if ((1 << cwl) > NT || (NT % (1 << cwl)) != 0) break;
      napass = vcv_cdiv(BM << cwl, NT);
    } else {
      const int jl = ilog2_ceil(tg.JMAX);
      if ((1 << jl) > NT || (NT % (1 << jl)) != 0) break;
      napass = vcv_cdiv((bkc * BM) << jl, NT);
    }
    const int nxpass = vcv_cdiv(bkc * xw, NT);
    if (napass > APT || (nxpass > XPT && !(allow_sync && bkc == 2))) break;
    best = bkc;
  }
  if (best == 0) return VCV_ENOFIT;
  tg.BKC = best;
  if (a.a_mode == 0) {
    tg.cw_log = ilog2_ceil(tg.BKC * a.K);
    if (tg.cw_log < min_cwl) tg.cw_log = min_cwl;
    tg.napass = vcv_cdiv(BM << tg.cw_log, NT);
    // a pass may overshoot BM rows when NT is not a power of two: the pitch absorbs the overshoot
    const int rows_touched = tg.napass * (NT >> tg.cw_log);
    tg.BMP = (rows_touched > BM ? rows_touched : BM) + 1;
    if ((tg.BMP & 1) == 0) tg.BMP += 1;
    tg.a_floats = tg.BKC * tg.JMAX * tg.BMP;
  } else {
    tg.cw_log = ilog2_ceil(tg.JMAX);
    tg.napass = vcv_cdiv((tg.BKC * BM) << tg.cw_log, NT);
    int pad = 32 >> tg.cw_log;
    if (pad < 1) pad = 1;
    tg.BMP = BM + pad;
    const int cl_alloc = vcv_cdiv(tg.napass * (NT >> tg.cw_log), BM);  // channels the passes touch
    tg.a_floats = (cl_alloc > tg.BKC ? cl_alloc : tg.BKC) * tg.JMAX * tg.BMP;
  }
  tg.nxpass = vcv_cdiv(tg.BKC * xw, NT);
  tg.xsync = tg.nxpass > XPT ? 1 : 0;
  const int x_floats = tg.nxpass * NT > tg.BKC * xw ? tg.nxpass * NT : tg.BKC * xw;
  const int U = a.Q * a.P;
  tg.ntu = vcv_cdiv(U, BN);
  tg.nmt = vcv_cdiv(a.Mg, BM);
  const size_t lds = ((size_t)tg.a_floats + (size_t)x_floats) * sizeof(float);
  if (lds > VCV_LDS_LIMIT) return VCV_ENOFIT;
  // 32-bit offset preconditions of the buffer-load staging
  if ((long long)a.Tin * a.P * 4 >= (1ll << 31) || (long long)a.Mg * a.Cg * a.K * 4 >= (1ll << 31) ||
      (long long)a.Mg * a.Tout * a.P >= (1ll << 31))
    return VCV_EINVAL;
  void (*kern)(const VcvConvArgs, const TileGeom);
  switch (a.in_tf) {
    case VCV_TF_NONE: kern = conv_gemm_kernel<TM, TN, WM, WN, INTF_NONE>; break;
    case VCV_TF_LEAKY: kern = conv_gemm_kernel<TM, TN, WM, WN, INTF_LEAKY>; break;
    case VCV_TF_DLEAKY: kern = conv_gemm_kernel<TM, TN, WM, WN, INTF_DLEAKY>; break;
    default: kern = conv_gemm_kernel<TM, TN, WM, WN, INTF_AUX>; break;
  }
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return VCV_EHIP;
  }
  dim3 grid(a.B * tg.ntu, a.G * tg.nmt, phases), block(NT);
  const double flops = 2.0 * a.B * a.G * a.Mg * a.Cg * a.K * a.P * (double)(phases > 1 ? a.Tin : a.Q);
  const int tag[12] = {a.B, a.G, a.Cg, a.Mg, a.K, a.Q, a.P, a.s, phases, a.a_mode, BM * 1000 + BN, tg.BKC};
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_CONV, flops, tag, 12, &ev0, &ev1);
  VCV_LAUNCH_EV(kern, grid, block, (unsigned)lds, st, ev0, ev1, a, tg);
  return vcv_check_launch();
}

}  // namespace

extern "C" int vcv_conv_gemm(const VcvConvArgs* args, void* stream) {
  if (!args) return VCV_EINVAL;
  const VcvConvArgs& a = *args;
  if (a.io != 0 || a.post_scale != 0.f || a.ms > 1) return VCV_EINVAL;  // (bf16 activations / post-scale: vcv_conv_bf16io_* only)
  if (a.B <= 0 || a.G <= 0 || a.Cg <= 0 || a.Mg <= 0 || a.Tin <= 0 || a.Tout <= 0 || a.P <= 0 ||
      a.K <= 0 || a.Q <= 0 || a.s <= 0)
    return VCV_EINVAL;
  if (a.in_tf >= VCV_TF_DLEAKY && !a.xaux) return VCV_EINVAL;
  if (a.out_tf != VCV_TF_NONE && !a.oaux) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int U = a.Q * a.P;
  const int phases = a.phases > 1 ? a.phases : 1;
  // tile choice: widest tile that still leaves >= ~2 workgroups per CU and is not mostly padding;
  // a tile whose staging does not fit the per-thread prefetch budget falls through to a narrower one
  const long long min_blocks = 512;
  auto ok = [&](int bm, int bn) {
    const int u32 = vcv_cdiv(U, 32) * 32;
    if (bn >= 2 * u32) return false;
    return (long long)a.B * vcv_cdiv(U, bn) * a.G * vcv_cdiv(a.Mg, bm) * phases >= min_blocks;
  };
  int rc = VCV_ENOFIT;
  // period-discriminator rows (160 < H*P <= 224 positions per batch element): one 7-wave tile covers a
  // whole row (no second, mostly empty 128-wide tile) and amortises the weight staging over 224 columns
  if (U > 160 && U <= 224 && a.Mg >= 64) {
    if (a.Mg >= 128 && (long long)a.B * a.G * vcv_cdiv(a.Mg, 128) * phases >= 224) rc = launch_conv<4, 1, 1, 7>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_conv<2, 1, 1, 7>(a, st);
    if (rc != VCV_ENOFIT) return rc;
  }
  if (a.Mg > 64) {
    if (ok(128, 128)) rc = launch_conv<2, 2, 2, 2>(a, st);
    if (rc == VCV_ENOFIT && ok(128, 64)) rc = launch_conv<2, 1, 2, 2>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 2, 2>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 2, 2>(a, st, true);
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 1, 4>(a, st);  // very long taps: only 32-row tiles fit the prefetch
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 1, 2>(a, st, true);
    return rc == VCV_ENOFIT ? VCV_EINVAL : rc;
  }
  if (a.Mg > 32) {
    // long sequences: 64 x 256 tile (2x2 accumulators per wave like the 128 x 128 tile)
    if (ok(64, 256)) rc = launch_conv<2, 2, 1, 4>(a, st);
    if (rc == VCV_ENOFIT && ok(64, 128)) rc = launch_conv<1, 2, 2, 2>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 2, 2>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 2, 2>(a, st, true);
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 1, 4>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 1, 2>(a, st, true);
    return rc == VCV_ENOFIT ? VCV_EINVAL : rc;
  }
  if (ok(32, 512)) rc = launch_conv<1, 4, 1, 4>(a, st);
  if (rc == VCV_ENOFIT && ok(32, 256)) rc = launch_conv<1, 2, 1, 4>(a, st);
  if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 1, 4>(a, st);
  if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 1, 2>(a, st);
  if (rc == VCV_ENOFIT) rc = launch_conv<1, 1, 1, 2>(a, st, true);
  return rc == VCV_ENOFIT ? VCV_EINVAL : rc;
}

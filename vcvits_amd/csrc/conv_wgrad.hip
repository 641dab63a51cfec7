// conv_wgrad.hip -- weight gradients of the conv family on the gfx950 fp32 matrix cores.
//
//   dw[m, c, k] += alpha * sum_{b, q, p} A[b, m, q, p] * Bsh[b, c, q*s + k*dj + off, p]
//
// GEMM view: M = channels of the un-shifted operand, N = (c, k) pairs in the weight's own memory
// order, reduction over the (b, q, p) positions.  A workgroup owns one BM x BN tile of dw and a
// strided subset of the position chunks (split-K over the grid's z axis, combined with fp32
// atomics that hit 128-B contiguous runs of dw).  Per chunk of BU positions it stages
//   As[u][m]        the un-shifted operand, transposed through LDS (k-major for the A fragment)
//   Xs[c][span]     one contiguous span per channel of the shifted operand -- all K taps read it
//                   at their own offset, so the shifted operand is not re-fetched per tap
//   tab[u]          the LDS offset of position u inside a staged span (handles P > 1 rows)
// and feeds v_mfma_f32_32x32x2_f32 with lane half h taking position 2i+h.
#include <cstdlib>

#include "common.h"
#include "prof.h"

namespace {

constexpr int VCV_ENOFIT = -100;
// positions per stage: 64 for plain sequences (P == 1), 32 for period layouts whose spans are row-wide
constexpr int WAPT = 32;  // un-shifted operand elements prefetched per thread per stage
constexpr int WXPT = 28;  // shifted operand elements prefetched per thread per stage

struct WgradGeom {
  int NCH, nmt, nnt, Z, nchunk_u, xw_log, napass, nxpass, a_floats, x_floats;
  int xpitch;  // LDS pitch of one staged channel span: 2^xw_log + skew, == K*dj*P (mod 32) so that the
               // (channel, tap) columns of one B fragment fall on distinct banks
  int xsync;  // 1: shifted-operand spans exceed the prefetch registers -> staged synchronously
};

__device__ __forceinline__ float ld_buf(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, 0, 0));
}

__device__ __forceinline__ float apply_tf(float v, float av, int tf, float slope) {
  if (tf == VCV_TF_LEAKY) return vcv_leaky(v, slope);
  if (tf == VCV_TF_DLEAKY) return v * vcv_dleaky(av, slope);
  if (tf == VCV_TF_DRELU) return av > 0.f ? v : 0.f;
  if (tf == VCV_TF_DTANH) return v * (1.f - av * av);
  if (tf == VCV_TF_DLOGCLAMP) return av > logf(slope) ? v * expf(-av) : 0.f;
  return v;
}

// Staging reads are buffer loads whose descriptor range check zero-fills rows past the channel
// count, positions past the sequence end and the convolution's zero padding.
template <int TM, int TN, int WM, int WN, bool AAUX, bool BAUX, int BU>
__global__ void __launch_bounds__(64 * WM * WN, 2)
conv_wgrad_kernel(const VcvWgradArgs p, const WgradGeom tg) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN, NT = 64 * NW;
  constexpr int BMP = BM + 1;
  constexpr int ARSTEP = NT / BU;
  constexpr int NAP = (BM * BU) / NT;  // un-shifted operand elements per thread per stage (exact)
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int nt = blockIdx.x;
  const int g = blockIdx.y / tg.nmt, mt = blockIdx.y % tg.nmt;
  const int z = blockIdx.z;
  const int K = p.K, Cg = p.Cg, Mg = p.Mg, P = p.P;
  const int N = Cg * K;
  const int n0 = nt * BN, m0 = mt * BM;
  const int cfirst = n0 / K;
  const int XW = 1 << tg.xw_log;

  float* As = smem;                  // [BU][BMP] (+ overshoot rows)
  float* Xs = As + tg.a_floats;      // [NCH][XW]
  int* tab = (int*)(Xs + tg.x_floats);  // [BU]

  int nofs[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int n = n0 + (wn * TN + tn) * 32 + l31;
    if (n > N - 1) n = N - 1;
    const int c = n / K, kw = n - c * K;
    nofs[tn] = (c - cfirst) * tg.xpitch + kw * p.dj * P;
  }

  const int jspan = (K - 1) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0;
  const long long U = (long long)p.Ta * P;
  const long long TbP = (long long)p.Tb * P;
  const unsigned urec = (unsigned)(U * 4), xrec = (unsigned)(TbP * 4);
  const int total = p.B * tg.nchunk_u;
  const int a_ul = tid & (BU - 1), a_row0 = tid / BU;
  const int rows_valid = Mg - m0 < BM ? Mg - m0 : BM;
  const int nch_valid = Cg - cfirst;  // channels of this tile that exist

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  float areg[NAP], aareg[AAUX ? NAP : 1];
  float xreg[WXPT], xareg[BAUX ? WXPT : 1];
  int cur_tab = 0, cur_f0 = 0;
  size_t cur_xbase = 0;

  auto load_chunk = [&](int ch) {
    const int b = ch / tg.nchunk_u;
    const int uc0 = (ch - b * tg.nchunk_u) * BU;
    const int qa = uc0 / P;
    const int rlo = qa * p.s + p.off + jmin;
    const int f0 = rlo * P;
    // un-shifted operand: one descriptor per row pass would cost SALU; rows are U floats apart, so a
    // single descriptor over the rows_valid rows of this batch element + a position check suffices
    const size_t abase = ((size_t)b * p.G * Mg + (size_t)g * Mg + m0) * (size_t)U;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + abase), 0, (int)((unsigned)rows_valid * urec), 0x00020000);
    __amdgpu_buffer_rsrc_t raa = ra;
    if (AAUX) raa = __builtin_amdgcn_make_buffer_rsrc((void*)(p.aaux + abase), 0, (int)((unsigned)rows_valid * urec), 0x00020000);
    const long long u = (long long)uc0 + a_ul;
    const unsigned av0 = u < U ? ((unsigned)a_row0 * (unsigned)U + (unsigned)u) * 4u : 0xFFFFFFFFu;
    const unsigned avstep = (unsigned)ARSTEP * (unsigned)U * 4u;
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
      const unsigned v = av0 == 0xFFFFFFFFu ? av0 : av0 + (unsigned)i * avstep;
      areg[i] = ld_buf(ra, v);
      if (AAUX) aareg[i] = ld_buf(raa, v);
    }
    const size_t xbase = ((size_t)b * p.G * Cg + (size_t)g * Cg + cfirst) * (size_t)TbP;
    cur_f0 = f0;
    cur_xbase = xbase;
    if (!tg.xsync)
#pragma unroll
    for (int i = 0; i < WXPT; ++i)
      if (i < tg.nxpass) {
        const int f = tid + i * NT;
        const int cl = __builtin_amdgcn_readfirstlane(f >> tg.xw_log);
        const int col = f & (XW - 1);
        const unsigned rec = cl < nch_valid ? xrec : 0u;
        const unsigned voff = (unsigned)(f0 + col) * 4u;
        __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + xbase + (size_t)cl * (size_t)TbP), 0, (int)rec, 0x00020000);
        xreg[i] = ld_buf(rx, voff);
        if (BAUX) {
          __amdgpu_buffer_rsrc_t rxa = __builtin_amdgcn_make_buffer_rsrc((void*)(p.baux + xbase + (size_t)cl * (size_t)TbP), 0, (int)rec, 0x00020000);
          xareg[i] = ld_buf(rxa, voff);
        }
      }
    // LDS offset of this thread's position inside a staged span (threads < BU fill the table)
    int t = 0;
    const long long ut = (long long)uc0 + tid;
    if (tid < BU && ut < U) {
      const int q = (int)(ut / P), pc = (int)(ut - (long long)q * P);
      t = ((q - qa) * p.s - jmin) * P + pc;
    }
    cur_tab = t;
  };

  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NAP; ++i)
      As[a_ul * BMP + a_row0 + i * ARSTEP] = apply_tf(areg[i], AAUX ? aareg[i] : 0.f, p.a_tf, p.slope);
    if (tg.xsync) {
      const int nel = tg.NCH << tg.xw_log;
      for (int f = tid; f < nel; f += NT) {
        const int cl = f >> tg.xw_log, col = f & (XW - 1);
        const long long ff = (long long)cur_f0 + col;
        float v = 0.f;
        if (cl < nch_valid && ff >= 0 && ff < TbP) {
          const size_t gi = cur_xbase + (size_t)cl * (size_t)TbP + (size_t)ff;
          v = vcv_tf(p.b[gi], p.b_tf, p.baux, gi, p.slope);
        }
        Xs[cl * tg.xpitch + col] = v;
      }
    } else
#pragma unroll
    for (int i = 0; i < WXPT; ++i)
      if (i < tg.nxpass) {
        const int f = tid + i * NT;
        Xs[(f >> tg.xw_log) * tg.xpitch + (f & (XW - 1))] = apply_tf(xreg[i], BAUX ? xareg[i] : 0.f, p.b_tf, p.slope);
      }
    if (tid < BU) tab[tid] = cur_tab;
  };

  if (z < total) {
    load_chunk(z);
    store_chunk();
    __syncthreads();
    for (int ch = z; ch < total; ch += tg.Z) {
      const bool more = ch + tg.Z < total;
      if (more) load_chunk(ch + tg.Z);
#pragma unroll 4
      for (int i = 0; i < BU; i += 2) {
        const int ul = i + h;
        const int bofs = tab[ul];
        float a[TM], bb[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = As[ul * BMP + (wm * TM + tm) * 32 + l31];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bb[tn] = Xs[nofs[tn] + bofs];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      }
      if (more) {
        __syncthreads();
        store_chunk();
        __syncthreads();
      }
    }
  }

#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + (wn * TN + tn) * 32 + l31;
    if (n >= N) continue;
    const int c = n / K, kw = n - c * K;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = m0 + (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (ml >= Mg) continue;
        size_t idx;
        if (p.transpose_out) idx = ((size_t)(g * Cg + c) * Mg + ml) * K + kw;
        else idx = ((size_t)(g * Mg + ml) * Cg) * K + n;
        unsafeAtomicAdd(p.dw + idx, p.alpha * acc[tm][tn][e]);
      }
    }
  }
}

inline int ilog2_ceil(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

template <int TM, int TN, int WM, int WN, int BU>
int launch_wgrad_bu(const VcvWgradArgs& a, hipStream_t st, bool allow_sync) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
  WgradGeom tg;
  const int N = a.Cg * a.K;
  tg.nnt = vcv_cdiv(N, BN);
  tg.nmt = vcv_cdiv(a.Mg, BM);
  tg.NCH = (BN - 1) / a.K + 2;
  if (tg.NCH > a.Cg + 1) tg.NCH = a.Cg + 1;
  const int qspan = (BU - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  const int rowmax = (qspan * a.s + (a.K - 1) * adj + 1) * a.P;
  tg.xw_log = ilog2_ceil(rowmax);
  if (tg.xw_log < 6) tg.xw_log = 6;
  // a power-of-two pitch would put the (channel, tap) columns of a B fragment on few LDS banks when
  // K is a multiple of the bank period; the pitch stays 2^k (needed by the shift decode) and the
  // conflicts that remain are paid in the LDS pipe, which has slack next to the 64-cycle MFMAs
  tg.napass = vcv_cdiv(BM * BU, NT);
  tg.nxpass = vcv_cdiv(tg.NCH << tg.xw_log, NT);
  if (tg.napass > WAPT) return VCV_ENOFIT;
  tg.xsync = tg.nxpass > WXPT ? 1 : 0;
  if (tg.xsync && !allow_sync) return VCV_ENOFIT;
  const int arows = tg.napass * (NT / BU);  // rows the passes touch (>= BM)
  tg.a_floats = BU * (BM + 1) + (arows > BM ? arows - BM : 0);
  {
    int skew = ((a.K * adj * a.P) % 32 + 32 - ((1 << tg.xw_log) % 32)) % 32;
    tg.xpitch = (1 << tg.xw_log) + skew;
  }
  const int xrows = vcv_cdiv(tg.nxpass * NT, 1 << tg.xw_log) > tg.NCH ? vcv_cdiv(tg.nxpass * NT, 1 << tg.xw_log) : tg.NCH;
  tg.x_floats = xrows * tg.xpitch;
  const long long U = (long long)a.Ta * a.P;
  if (U * a.Mg * 4 >= (1ll << 31) || (long long)a.Tb * a.P * 4 >= (1ll << 31)) return VCV_EINVAL;
  tg.nchunk_u = (int)((U + BU - 1) / BU);
  const long long total = (long long)a.B * tg.nchunk_u;
  long long tiles = (long long)tg.nnt * tg.nmt * a.G;
  long long Z = 1024 / tiles;
  if (Z < 1) Z = 1;
  if (Z > total) Z = total;
  if (vcv_get_deterministic()) Z = 1;  // (the position splits meet in fp32 atomics)
  tg.Z = (int)Z;
  const size_t lds = ((size_t)tg.a_floats + (size_t)tg.x_floats + BU) * sizeof(float);
  if (lds > VCV_LDS_LIMIT) return VCV_ENOFIT;
  const bool aaux = a.a_tf >= VCV_TF_DLEAKY, baux = a.b_tf >= VCV_TF_DLEAKY;
  if (aaux && baux) return VCV_EINVAL;
  auto kern = aaux ? conv_wgrad_kernel<TM, TN, WM, WN, true, false, BU>
                   : (baux ? conv_wgrad_kernel<TM, TN, WM, WN, false, true, BU>
                           : conv_wgrad_kernel<TM, TN, WM, WN, false, false, BU>);
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return VCV_EHIP;
  }
  dim3 grid(tg.nnt, a.G * tg.nmt, tg.Z), block(NT);
  const double flops = 2.0 * a.B * a.G * a.Mg * a.Cg * a.K * a.P * (double)a.Ta;
  const int tag[12] = {a.B, a.G, a.Cg, a.Mg, a.K, a.Ta, a.P, a.s, tg.Z, 0, BM * 1000 + BN, tg.NCH};
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_WGRAD, flops, tag, 12, &ev0, &ev1);
  VCV_LAUNCH_EV(kern, grid, block, (unsigned)lds, st, ev0, ev1, a, tg);
  return vcv_check_launch();
}

template <int TM, int TN, int WM, int WN>
int launch_wgrad(const VcvWgradArgs& a, hipStream_t st, bool allow_sync = false) {
  if (a.P == 1 && a.a_tf < VCV_TF_DLEAKY) {
    const int rc = launch_wgrad_bu<TM, TN, WM, WN, 64>(a, st, false);
    if (rc != VCV_ENOFIT) return rc;
  }
  return launch_wgrad_bu<TM, TN, WM, WN, 32>(a, st, allow_sync);
}

__device__ __forceinline__ float bias_term(float v, float y, int tf, float slope) {
  if (tf == VCV_TF_DLEAKY) return v * vcv_dleaky(y, slope);
  if (tf == VCV_TF_DRELU) return y > 0.f ? v : 0.f;
  if (tf == VCV_TF_DTANH) return v * (1.f - y * y);
  return v;
}

// db[c] += sum over (b, t) of dy[b, c, t] (optionally times the activation derivative at aux).  Work units are
// 1024-float pieces of the contiguous (b, c) rows; VEC = rows are 16-byte aligned (T % 4 == 0).
template <bool VEC>
__global__ void __launch_bounds__(256)
bias_grad_kernel(const float* __restrict__ dy, const float* __restrict__ aux, float* __restrict__ db,
                 int B, int C, int T, int tf, float slope, int nseg, int acc) {
  const int c = blockIdx.x, seg = blockIdx.y;
  const int nchunk = (T + 1023) >> 10;
  const int units = B * nchunk;
  const int per = (units + nseg - 1) / nseg;
  const int lo = seg * per;
  const int hi = lo + per < units ? lo + per : units;
  float s = 0.f;
  for (int unit = lo; unit < hi; ++unit) {
    const int b = unit / nchunk, ch = unit - b * nchunk;
    const size_t row = ((size_t)b * C + c) * (size_t)T;
    if (VEC) {
      const int t = (ch << 10) + threadIdx.x * 4;
      if (t < T) {
        const float4 v = *reinterpret_cast<const float4*>(dy + row + t);
        if (tf >= VCV_TF_DLEAKY) {
          const float4 y = *reinterpret_cast<const float4*>(aux + row + t);
          s += bias_term(v.x, y.x, tf, slope) + bias_term(v.y, y.y, tf, slope) + bias_term(v.z, y.z, tf, slope) +
               bias_term(v.w, y.w, tf, slope);
        } else {
          s += (v.x + v.y) + (v.z + v.w);
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int t = (ch << 10) + k * 256 + threadIdx.x;
        if (t < T) s += bias_term(dy[row + t], tf >= VCV_TF_DLEAKY ? aux[row + t] : 0.f, tf, slope);
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = red[0] + red[1] + red[2] + red[3];
    if (nseg == 1 && !acc) db[c] = r; else unsafeAtomicAdd(db + c, r);
  }
}

}  // namespace

int vcv_wgrad_dma_try(const VcvWgradArgs& a, hipStream_t st);  // wgrad_dma.hip

extern "C" int vcv_conv_wgrad(const VcvWgradArgs* args, void* stream) {
  if (!args) return VCV_EINVAL;
  const VcvWgradArgs& a = *args;
  if (a.B <= 0 || a.G <= 0 || a.Cg <= 0 || a.Mg <= 0 || a.Ta <= 0 || a.Tb <= 0 || a.P <= 0 ||
      a.K <= 0 || a.s <= 0)
    return VCV_EINVAL;
  if (a.a_tf >= VCV_TF_DLEAKY && !a.aaux) return VCV_EINVAL;
  if (a.b_tf >= VCV_TF_DLEAKY && !a.baux) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int N = a.Cg * a.K;
  const bool use_dma = vcv_tuning().wgrad_dma != 0;
  if (a.dbias && (a.a_tf != VCV_TF_NONE || a.G != 1)) return VCV_EINVAL;
  if (use_dma) {
    const int rcd = vcv_wgrad_dma_try(a, st);
    if (rcd != VCV_ENOFIT) return rcd;
  }
  if (a.dbias) {
    // the register-staged kernel does not collect the row sums: one streaming pass over `a`
    const int rb = vcv_bias_grad(a.a, nullptr, a.dbias, a.B, a.Mg, a.Ta * a.P, VCV_TF_NONE, a.slope, 1, stream);
    if (rb != VCV_OK) return rb;
  }
  int rc = VCV_ENOFIT;
  if (a.Mg > 64) {
    if (N > 64) rc = launch_wgrad<2, 2, 2, 2>(a, st);
    if (rc == VCV_ENOFIT && N > 32) rc = launch_wgrad<2, 1, 2, 2>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_wgrad<1, 1, 4, 1>(a, st);
  } else if (a.Mg > 32) {
    if (N > 64) rc = launch_wgrad<1, 2, 2, 2>(a, st);
    if (rc == VCV_ENOFIT && N > 32) rc = launch_wgrad<1, 1, 2, 2>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_wgrad<1, 1, 2, 1>(a, st);
  } else {
    if (N > 128) rc = launch_wgrad<1, 2, 1, 4>(a, st);
    if (rc == VCV_ENOFIT && N > 64) rc = launch_wgrad<1, 1, 1, 4>(a, st);
    if (rc == VCV_ENOFIT && N > 32) rc = launch_wgrad<1, 1, 1, 2>(a, st);
    if (rc == VCV_ENOFIT) rc = launch_wgrad<1, 1, 1, 1>(a, st);
  }
  if (rc == VCV_ENOFIT) {  // no tile fits the prefetch registers: synchronous staging of the spans
    if (a.Mg > 64) rc = launch_wgrad<1, 1, 4, 1>(a, st, true);
    else if (a.Mg > 32) rc = launch_wgrad<1, 1, 2, 1>(a, st, true);
    else rc = launch_wgrad<1, 1, 1, 1>(a, st, true);
  }
  return rc == VCV_ENOFIT ? VCV_EINVAL : rc;
}

// Short rows (T <= 1024, T % 4 == 0: the 1 x 1 convs of the encoders / flow at 200-400 frames): the kernel above gives a
// (b, c) row to the whole workgroup per iteration -- 51-96 of its 256 threads load, B iterations one after the other
// (17 us for 9 MB).  Here the (b, t / 4) items of a channel are dealt to the threads flat, eight 16-byte loads in flight each.
template <bool VEC>
__global__ void __launch_bounds__(256)
bias_grad_rows_kernel(const float* __restrict__ dy, const float* __restrict__ aux, float* __restrict__ db, int B, int C, int T,
                      int tf, float slope, int acc) {
  const int c = blockIdx.x;
  const int t4n = VEC ? T >> 2 : T, total = B * t4n;  // items of a channel: 16-byte pieces (VEC) or single elements
  float s = 0.f;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * 8) {
    float4 v[8], y[8];
    bool in[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256;
      in[u] = i < total;
      const int b = in[u] ? i / t4n : 0, t4 = in[u] ? i - b * t4n : 0;
      const size_t o = ((size_t)b * C + c) * (size_t)T + (VEC ? 4 * t4 : t4);
      if (VEC) {
        v[u] = *reinterpret_cast<const float4*>(dy + o);
        if (tf >= VCV_TF_DLEAKY) y[u] = *reinterpret_cast<const float4*>(aux + o);
      } else {
        v[u].x = dy[o];
        if (tf >= VCV_TF_DLEAKY) y[u].x = aux[o];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (in[u]) {
        if (VEC) {
          if (tf >= VCV_TF_DLEAKY)
            s += bias_term(v[u].x, y[u].x, tf, slope) + bias_term(v[u].y, y[u].y, tf, slope) + bias_term(v[u].z, y[u].z, tf, slope) +
                 bias_term(v[u].w, y[u].w, tf, slope);
          else
            s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
        } else {
          s += bias_term(v[u].x, tf >= VCV_TF_DLEAKY ? y[u].x : 0.f, tf, slope);
        }
      }
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = red[0] + red[1] + red[2] + red[3];
    if (!acc) db[c] = r; else db[c] += r;  // (one workgroup per channel: no atomic, deterministic)
  }
}

extern "C" int vcv_bias_grad(const float* dy, const float* aux, float* dbias, int B, int C, int T,
                             int tf, float slope, int accumulate, void* stream) {
  if (!dy || !dbias || B <= 0 || C <= 0 || T <= 0) return VCV_EINVAL;
  if (tf >= VCV_TF_DLEAKY && !aux) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool rows_on = vcv_tuning().bias_rows != 0;
  if (rows_on && T <= 1024 && T % 4 == 0 && (long long)B * (T / 4) <= 256 * 64) {
    hipLaunchKernelGGL(bias_grad_rows_kernel<true>, dim3(C), dim3(256), 0, st, dy, aux, dbias, B, C, T, tf, slope, accumulate);
    return vcv_check_launch();
  }
  if (rows_on && T <= 1024 && (long long)B * T <= 256 * 64) {  // (rows of any length: element by element, still flat)
    hipLaunchKernelGGL(bias_grad_rows_kernel<false>, dim3(C), dim3(256), 0, st, dy, aux, dbias, B, C, T, tf, slope, accumulate);
    return vcv_check_launch();
  }
  // enough workgroups to fill the chip, each with at least ~8 pieces of 1024 floats
  const long long units = (long long)B * ((T + 1023) / 1024);
  long long nseg = (1024 + C - 1) / C;
  if (nseg > units / 8) nseg = units / 8;
  if (nseg < 1 || vcv_get_deterministic()) nseg = 1;
  if (nseg > 1 && !accumulate && vcv_zero_async(dbias, sizeof(float) * C, st) != hipSuccess) return VCV_EHIP;
  if (T % 4 == 0)
    hipLaunchKernelGGL(bias_grad_kernel<true>, dim3(C, (unsigned)nseg), dim3(256), 0, st, dy, aux, dbias, B, C, T, tf,
                       slope, (int)nseg, accumulate);
  else
    hipLaunchKernelGGL(bias_grad_kernel<false>, dim3(C, (unsigned)nseg), dim3(256), 0, st, dy, aux, dbias, B, C, T, tf,
                       slope, (int)nseg, accumulate);
  return vcv_check_launch();
}

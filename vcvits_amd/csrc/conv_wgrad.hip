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
#include "common.h"
#include "prof.h"

namespace {

constexpr int VCV_ENOFIT = -100;
constexpr int BU = 32;    // positions per stage (power of two)
constexpr int WAPT = 16;  // un-shifted operand elements prefetched per thread per stage
constexpr int WXPT = 16;  // shifted operand elements prefetched per thread per stage

struct WgradGeom {
  int NCH, ROWP, nmt, nnt, Z, nchunk_u, xw_log, napass, nxpass;
  int xsync;  // 1: the shifted-operand spans exceed the prefetch registers -> staged synchronously
};

template <int TM, int TN, int WM, int WN, bool AAUX, bool BAUX>
__global__ void __launch_bounds__(64 * WM * WN, 2)
conv_wgrad_kernel(const VcvWgradArgs p, const WgradGeom tg) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN, NT = 64 * NW;
  constexpr int BMP = BM + 1;
  constexpr int ARSTEP = NT / BU;
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
  const int ROWP = tg.ROWP, NCH = tg.NCH;

  float* As = smem;                    // [BU][BMP]
  float* Xs = As + BU * BMP;           // [NCH][ROWP]
  int* tab = (int*)(Xs + NCH * ROWP);  // [BU]

  int nofs[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int n = n0 + (wn * TN + tn) * 32 + l31;
    if (n > N - 1) n = N - 1;
    const int c = n / K, kw = n - c * K;
    nofs[tn] = (c - cfirst) * ROWP + kw * p.dj * P;
  }

  const int jspan = (K - 1) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0, jmax = jspan > 0 ? jspan : 0;
  const long long U = (long long)p.Ta * P;
  const long long TbP = (long long)p.Tb * P;
  const int total = p.B * tg.nchunk_u;
  const int xw = 1 << tg.xw_log;
  const int a_ul = tid & (BU - 1), a_row0 = tid / BU;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  float areg[WAPT], aareg[AAUX ? WAPT : 1];
  float xreg[WXPT], xareg[BAUX ? WXPT : 1];
  int cur_rowlen = 0, cur_tab = 0;
  long long cur_f0 = 0;
  size_t cur_xoff = 0;

  auto load_chunk = [&](int ch) {
    const int b = ch / tg.nchunk_u;
    const int uc0 = (ch - b * tg.nchunk_u) * BU;
    const int qa = uc0 / P;
    int qb = (uc0 + BU - 1) / P;
    if (qb > p.Ta - 1) qb = p.Ta - 1;
    const int rlo = qa * p.s + p.off + jmin;
    cur_rowlen = ((qb - qa) * p.s + (jmax - jmin) + 1) * P;
    const long long f0 = (long long)rlo * P;
    // un-shifted operand: rows m, positions uc0 + a_ul
    const long long u = (long long)uc0 + a_ul;
    const float* ab = p.a + ((size_t)b * p.G * Mg + (size_t)g * Mg + m0) * (size_t)U;
    const float* aab = AAUX ? p.aaux + ((size_t)b * p.G * Mg + (size_t)g * Mg + m0) * (size_t)U : nullptr;
#pragma unroll
    for (int i = 0; i < WAPT; ++i) {
      const int row = a_row0 + i * ARSTEP;
      float v = 0.f, av = 0.f;
      if (i < tg.napass && row < BM && m0 + row < Mg && u < U) {
        const size_t gi = (size_t)row * (size_t)U + (size_t)u;
        v = ab[gi];
        if (AAUX) av = aab[gi];
      }
      areg[i] = v;
      if (AAUX) aareg[i] = av;
    }
    cur_f0 = f0;
    cur_xoff = ((size_t)b * p.G * Cg + (size_t)g * Cg + cfirst) * (size_t)TbP;
    const float* xb = p.b + cur_xoff;
    const float* xab = BAUX ? p.baux + cur_xoff : nullptr;
    if (!tg.xsync) {
#pragma unroll
    for (int i = 0; i < WXPT; ++i) {
      const int f = tid + i * NT;
      const int cl = f >> tg.xw_log, col = f & (xw - 1);
      float v = 0.f, av = 0.f;
      if (i < tg.nxpass && cl < NCH && col < cur_rowlen && cfirst + cl < Cg) {
        const long long ff = f0 + col;
        if (ff >= 0 && ff < TbP) {
          const size_t gi = (size_t)cl * (size_t)TbP + (size_t)ff;
          v = xb[gi];
          if (BAUX) av = xab[gi];
        }
      }
      xreg[i] = v;
      if (BAUX) xareg[i] = av;
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
    for (int i = 0; i < WAPT; ++i) {
      const int row = a_row0 + i * ARSTEP;
      if (i < tg.napass && row < BM) {
        float v = areg[i];
        if (p.a_tf == VCV_TF_LEAKY) v = vcv_leaky(v, p.slope);
        if (AAUX) {
          const float av = aareg[i];
          if (p.a_tf == VCV_TF_DLEAKY) v *= vcv_dleaky(av, p.slope);
          else if (p.a_tf == VCV_TF_DRELU) v = av > 0.f ? v : 0.f;
          else if (p.a_tf == VCV_TF_DTANH) v *= 1.f - av * av;
          else if (p.a_tf == VCV_TF_DLOGCLAMP) v = av > logf(p.slope) ? v * expf(-av) : 0.f;
        }
        As[a_ul * BMP + row] = v;
      }
    }
    if (tg.xsync) {
      // generic path for very wide spans (large period x stride): load + transform + store in one loop
      const int nel = NCH << tg.xw_log;
      for (int f = tid; f < nel; f += NT) {
        const int cl = f >> tg.xw_log, col = f & (xw - 1);
        if (col >= cur_rowlen) continue;
        float v = 0.f;
        const long long ff = cur_f0 + col;
        if (cfirst + cl < Cg && ff >= 0 && ff < TbP) {
          const size_t gi = cur_xoff + (size_t)cl * (size_t)TbP + (size_t)ff;
          v = vcv_tf(p.b[gi], p.b_tf, p.baux, gi, p.slope);
        }
        Xs[cl * ROWP + col] = v;
      }
    } else {
#pragma unroll
    for (int i = 0; i < WXPT; ++i) {
      const int f = tid + i * NT;
      const int cl = f >> tg.xw_log, col = f & (xw - 1);
      if (i < tg.nxpass && cl < NCH && col < cur_rowlen) {
        float v = xreg[i];
        if (p.b_tf == VCV_TF_LEAKY) v = vcv_leaky(v, p.slope);
        if (BAUX) {
          const float av = xareg[i];
          if (p.b_tf == VCV_TF_DLEAKY) v *= vcv_dleaky(av, p.slope);
          else if (p.b_tf == VCV_TF_DRELU) v = av > 0.f ? v : 0.f;
          else if (p.b_tf == VCV_TF_DTANH) v *= 1.f - av * av;
          else if (p.b_tf == VCV_TF_DLOGCLAMP) v = av > logf(p.slope) ? v * expf(-av) : 0.f;
        }
        Xs[cl * ROWP + col] = v;
      }
    }
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

template <int TM, int TN, int WM, int WN>
int launch_wgrad(const VcvWgradArgs& a, hipStream_t st, bool allow_sync = false) {
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
  tg.ROWP = rowmax | 1;  // odd pitch spreads the (channel, tap) columns of a B fragment over the banks
  tg.napass = vcv_cdiv(BM * BU, NT);
  tg.nxpass = vcv_cdiv(tg.NCH << tg.xw_log, NT);
  if (tg.napass > WAPT) return VCV_ENOFIT;
  tg.xsync = tg.nxpass > WXPT ? 1 : 0;
  if (tg.xsync && !allow_sync) return VCV_ENOFIT;
  const long long U = (long long)a.Ta * a.P;
  tg.nchunk_u = (int)((U + BU - 1) / BU);
  const long long total = (long long)a.B * tg.nchunk_u;
  long long tiles = (long long)tg.nnt * tg.nmt * a.G;
  long long Z = 1024 / tiles;
  if (Z < 1) Z = 1;
  if (Z > total) Z = total;
  tg.Z = (int)Z;
  const size_t lds = ((size_t)BU * (BM + 1) + (size_t)tg.NCH * tg.ROWP + BU) * sizeof(float);
  if (lds > VCV_LDS_LIMIT) return VCV_ENOFIT;
  const bool aaux = a.a_tf >= VCV_TF_DLEAKY, baux = a.b_tf >= VCV_TF_DLEAKY;
  if (aaux && baux) return VCV_EINVAL;
  auto kern = aaux ? conv_wgrad_kernel<TM, TN, WM, WN, true, false>
                   : (baux ? conv_wgrad_kernel<TM, TN, WM, WN, false, true>
                           : conv_wgrad_kernel<TM, TN, WM, WN, false, false>);
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return VCV_EHIP;
  }
  dim3 grid(tg.nnt, a.G * tg.nmt, tg.Z), block(NT);
  const double flops = 2.0 * a.B * a.G * a.Mg * a.Cg * a.K * a.P * (double)a.Ta;
  const int tag[12] = {a.B, a.G, a.Cg, a.Mg, a.K, a.Ta, a.P, a.s, tg.Z, tg.xsync, BM * 1000 + BN, tg.NCH};
  const int slot = vcv_prof_start(VCV_PROF_WGRAD, flops, st, tag, 12);
  hipLaunchKernelGGL(kern, grid, block, lds, st, a, tg);
  vcv_prof_stop(slot, st);
  return vcv_check_launch();
}

__global__ void __launch_bounds__(256)
bias_grad_kernel(const float* __restrict__ dy, const float* __restrict__ aux, float* __restrict__ db,
                 int B, int C, int T, int tf, float slope, int nseg) {
  const int c = blockIdx.x, seg = blockIdx.y;
  const long long total = (long long)B * T;
  const long long per = (total + nseg - 1) / nseg;
  const long long lo = seg * per;
  long long hi = lo + per;
  if (hi > total) hi = total;
  float s = 0.f;
  for (long long i = lo + threadIdx.x; i < hi; i += 256) {
    const long long b = i / T, t = i - b * T;
    const size_t idx = ((size_t)b * C + c) * T + t;
    float v = dy[idx];
    if (tf == VCV_TF_DLEAKY) v *= vcv_dleaky(aux[idx], slope);
    else if (tf == VCV_TF_DRELU) v = aux[idx] > 0.f ? v : 0.f;
    else if (tf == VCV_TF_DTANH) v *= 1.f - aux[idx] * aux[idx];
    s += v;
  }
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(db + c, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

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

extern "C" int vcv_bias_grad(const float* dy, const float* aux, float* dbias, int B, int C, int T,
                             int tf, float slope, void* stream) {
  if (!dy || !dbias || B <= 0 || C <= 0 || T <= 0) return VCV_EINVAL;
  if (tf >= VCV_TF_DLEAKY && !aux) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(dbias, 0, sizeof(float) * C, st) != hipSuccess) return VCV_EHIP;
  const long long total = (long long)B * T;
  int nseg = (int)((total + 16383) / 16384);
  if (nseg < 1) nseg = 1;
  if (nseg > 64) nseg = 64;
  hipLaunchKernelGGL(bias_grad_kernel, dim3(C, nseg), dim3(256), 0, st, dy, aux, dbias, B, C, T, tf,
                     slope, nseg);
  return vcv_check_launch();
}

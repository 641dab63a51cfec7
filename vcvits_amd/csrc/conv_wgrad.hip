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

struct WgradGeom {
  int BU, NCH, ROWP, nmt, nnt, Z, nchunk_u;
};

template <int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN)
conv_wgrad_kernel(const VcvWgradArgs p, const WgradGeom tg) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN, NT = 64 * NW;
  constexpr int BMP = BM + 1;
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
  const int BU = tg.BU, ROWP = tg.ROWP, NCH = tg.NCH;

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

  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  for (int ch = z; ch < total; ch += tg.Z) {
    const int b = ch / tg.nchunk_u;
    const int uc0 = (ch - b * tg.nchunk_u) * BU;
    const int qa = uc0 / P;
    int qb = (uc0 + BU - 1) / P;
    if (qb > p.Ta - 1) qb = p.Ta - 1;
    const int rlo = qa * p.s + p.off + jmin;
    const int rowlen = ((qb - qa) * p.s + (jmax - jmin) + 1) * P;
    const long long f0 = (long long)rlo * P;
    __syncthreads();
    // stage the un-shifted operand, transposed
    for (int row = wave; row < BM; row += NW) {
      const int m = m0 + row;
      const size_t base = ((size_t)b * p.G * Mg + (size_t)g * Mg + m) * (size_t)U;
      for (int ul = lane; ul < BU; ul += 64) {
        const long long u = (long long)uc0 + ul;
        float v = 0.f;
        if (m < Mg && u < U) {
          v = p.a[base + u];
          v = vcv_tf(v, p.a_tf, p.aaux, base + u, p.slope);
        }
        As[ul * BMP + row] = v;
      }
    }
    // stage the shifted operand spans
    for (int cl = wave; cl < NCH; cl += NW) {
      const int c = cfirst + cl;
      const size_t base = ((size_t)b * p.G * Cg + (size_t)g * Cg + c) * (size_t)TbP;
      float* xs = Xs + cl * ROWP;
      for (int i = lane; i < rowlen; i += 64) {
        const long long f = f0 + i;
        float v = 0.f;
        if (c < Cg && f >= 0 && f < TbP) {
          v = p.b[base + f];
          v = vcv_tf(v, p.b_tf, p.baux, base + f, p.slope);
        }
        xs[i] = v;
      }
    }
    for (int ul = tid; ul < BU; ul += NT) {
      const long long u = (long long)uc0 + ul;
      int t = 0;
      if (u < U) {
        const int q = (int)(u / P), pc = (int)(u - (long long)q * P);
        t = ((q - qa) * p.s - jmin) * P + pc;
      }
      tab[ul] = t;
    }
    __syncthreads();
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

template <int TM, int TN, int WM, int WN>
int launch_wgrad(const VcvWgradArgs& a, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  WgradGeom tg;
  tg.BU = 64;
  const int N = a.Cg * a.K;
  tg.nnt = vcv_cdiv(N, BN);
  tg.nmt = vcv_cdiv(a.Mg, BM);
  tg.NCH = (BN - 1) / a.K + 2;
  if (tg.NCH > a.Cg + 1) tg.NCH = a.Cg + 1;
  const int qspan = (tg.BU - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  tg.ROWP = (qspan * a.s + (a.K - 1) * adj + 1) * a.P;
  // odd pitch spreads the (channel, tap) columns of a B fragment over the LDS banks
  if ((tg.ROWP & 1) == 0) tg.ROWP += 1;
  const long long U = (long long)a.Ta * a.P;
  tg.nchunk_u = (int)((U + tg.BU - 1) / tg.BU);
  const long long total = (long long)a.B * tg.nchunk_u;
  long long tiles = (long long)tg.nnt * tg.nmt * a.G;
  long long Z = 1024 / tiles;
  if (Z < 1) Z = 1;
  if (Z > total) Z = total;
  tg.Z = (int)Z;
  const size_t lds = ((size_t)tg.BU * (BM + 1) + (size_t)tg.NCH * tg.ROWP + tg.BU) * sizeof(float);
  if (lds > VCV_LDS_LIMIT) return VCV_ELDS;
  auto kern = conv_wgrad_kernel<TM, TN, WM, WN>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return VCV_EHIP;
  }
  dim3 grid(tg.nnt, a.G * tg.nmt, tg.Z), block(64 * WM * WN);
  const double flops = 2.0 * a.B * a.G * a.Mg * a.Cg * a.K * a.P * (double)a.Ta;
  const int slot = vcv_prof_start(VCV_PROF_WGRAD, flops, st);
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
  if (a.Mg > 64) {
    if (N > 64) return launch_wgrad<2, 2, 2, 2>(a, st);
    return launch_wgrad<2, 1, 2, 2>(a, st);
  }
  if (a.Mg > 32) {
    if (N > 64) return launch_wgrad<1, 2, 2, 2>(a, st);
    return launch_wgrad<1, 1, 2, 2>(a, st);
  }
  if (N > 128) return launch_wgrad<1, 2, 1, 4>(a, st);
  return launch_wgrad<1, 1, 1, 4>(a, st);
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

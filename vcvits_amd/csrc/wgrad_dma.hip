// wgrad_dma.hip -- LDS-DMA variant of the MFMA weight-gradient kernel (see conv_wgrad.hip for the GEMM
// mapping: M = channels of the un-shifted operand, N = (c, k) in the weight's memory order, reduction over
// positions, split over the grid and combined with fp32 atomics).
//
// Both operands are activations, so nothing needs packing: per stage of 64 positions
//   As[m][u]   one `buffer_load_dword ... lds` per channel row (64 floats; rows past M and positions past
//              the sequence end are zero-filled by the descriptor's range check); the odd row pitch makes
//              the A-fragment read (lane -> consecutive m) conflict-free without a transpose pass;
//   Xs[c][span] the shifted operand's contiguous span per channel (all K taps read it at their offset);
// double-buffered: the DMA of stage s+1 runs under the MFMA loop of stage s, one barrier per stage, no
// staging registers.  Input leaky-ReLU (ResBlock / generator convs) is applied to the fragments as read.
#include "common.h"
#include "prof.h"

namespace {

constexpr int BU = 64, AP = BU + 1;
typedef __attribute__((address_space(3))) void* lds_ptr;

struct WgGeom {
  int NCH, nXrow, XP, nmt, nnt, Z, nchunk_u, buf_floats, a_floats;
};

template <int TM, int TN, int WM, int WN, bool LA, bool LB, bool BIAS>
__global__ void __launch_bounds__(64 * WM * WN)
wgrad_dma_kernel(const VcvWgradArgs p, const WgGeom tg) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN;
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;

  const int nt = blockIdx.x, mt = blockIdx.y, z = blockIdx.z;
  const int K = p.K, Cg = p.Cg, Mg = p.Mg, P = p.P;
  const int N = Cg * K;
  const int n0 = nt * BN, m0 = mt * BM;
  const int cfirst = n0 / K;
  const int XP = tg.XP, NCH = tg.NCH, nXrow = tg.nXrow;

  int nofs[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int n = n0 + (wn * TN + tn) * 32 + l31;
    if (n > N - 1) n = N - 1;
    const int c = n / K, kw = n - c * K;
    nofs[tn] = (c - cfirst) * XP + kw * p.dj * P;
  }
  const int jspan = (K - 1) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0;
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
  // bias gradient = row sums of the un-shifted operand: collected by the first column tile's first wave column from
  // the A fragments it reads anyway (p.dbias, see include/vcvits_hip.h)
  const bool do_bias = BIAS && nt == 0 && wn == 0;
  float bsum[TM];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) bsum[tm] = 0.f;

  auto issue = [&](int ch, int buf) {
    float* As = smem + buf * tg.buf_floats;
    float* Xs = As + tg.a_floats;
    int* tab = (int*)(Xs + NCH * XP);
    const int b = ch / tg.nchunk_u;
    const int uc0 = (ch - b * tg.nchunk_u) * BU;
    const int qa = uc0 / P;
    const int f0 = (qa * p.s + p.off + jmin) * P;
    const float* ab = p.a + ((size_t)b * Mg + m0) * (size_t)U;
    const unsigned avoff = (unsigned)(uc0 + lane) * 4u;
    for (int i = wave; i < BM; i += NW) {
      const unsigned rec = (m0 + i < Mg) ? (unsigned)(U * 4) : 0u;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(ab + (size_t)i * (size_t)U), 0, (int)rec, 0x00020000);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(As + i * AP), 4, avoff, 0, 0, 0);
    }
    const float* xb = p.b + ((size_t)b * Cg + cfirst) * (size_t)TbP;
    const int nX = NCH * nXrow;
    for (int i = wave; i < nX; i += NW) {
      const int cl = i / nXrow, part = i - cl * nXrow;
      const unsigned rec = (cfirst + cl < Cg) ? (unsigned)(TbP * 4) : 0u;
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)cl * (size_t)TbP), 0, (int)rec, 0x00020000);
      const unsigned voff = (unsigned)(f0 + part * 64 + lane) * 4u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr)(Xs + cl * XP + part * 64), 4, voff, 0, 0, 0);
    }
    if (tid < BU) {
      const long long u = (long long)uc0 + tid;
      int t = 0;
      if (u < U) {
        const int q = (int)(u / P), pc = (int)(u - (long long)q * P);
        t = ((q - qa) * p.s - jmin) * P + pc;
      }
      tab[tid] = t;
    }
  };

  if (z < total) {
    issue(z, 0);
    int bufi = 0;
    for (int ch = z; ch < total; ch += tg.Z) {
      __syncthreads();
      if (ch + tg.Z < total) issue(ch + tg.Z, bufi ^ 1);
      const float* As = smem + bufi * tg.buf_floats;
      const float* Xs = As + tg.a_floats;
      const int* tab = (const int*)(Xs + NCH * XP);
#pragma unroll 4
      for (int i = 0; i < BU; i += 2) {
        const int ul = i + h;
        const int bofs = tab[ul];
        float a[TM], bb[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          float v = As[((wm * TM + tm) * 32 + l31) * AP + ul];
          if (LA) v = fmaxf(v, v * p.slope);
          a[tm] = v;
          if (BIAS && do_bias) bsum[tm] += v;
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
          float v = Xs[nofs[tn] + bofs];
          if (LB) v = fmaxf(v, v * p.slope);
          bb[tn] = v;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      }
      bufi ^= 1;
    }
  }

  if (do_bias) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float s2 = bsum[tm] + __shfl_xor(bsum[tm], 32, 64);  // the two position parities of the lane halves
      const int ml = m0 + (wm * TM + tm) * 32 + l31;
      if (h == 0 && ml < Mg) unsafeAtomicAdd(p.dbias + ml, s2);
    }
  }
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = n0 + (wn * TN + tn) * 32 + l31;
    if (n >= N) continue;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ml = m0 + (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (ml >= Mg) continue;
        if (p.slab) p.slab[(size_t)z * Mg * N + (size_t)ml * N + n] = acc[tm][tn][e];
        else unsafeAtomicAdd(p.dw + (size_t)ml * N + n, p.alpha * acc[tm][tn][e]);
      }
    }
  }
}

// dw[i] += alpha * sum_z slab[z][i] in a fixed order (the deterministic combine of a split reduction)
__global__ void __launch_bounds__(256) wgrad_slab_finish_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                size_t n, int Z, float alpha) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = slab[i];
  for (int z = 1; z < Z; ++z) s += slab[(size_t)z * n + i];
  dw[i] += alpha * s;
}

template <int TM, int TN, int WM, int WN>
int launch(const VcvWgradArgs& a, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NT = 64 * WM * WN;
  WgGeom g;
  const int N = a.Cg * a.K;
  g.nnt = vcv_cdiv(N, BN);
  g.nmt = vcv_cdiv(a.Mg, BM);
  g.NCH = (BN - 1) / a.K + 2;
  if (g.NCH > a.Cg + 1) g.NCH = a.Cg + 1;
  const int qspan = (BU - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  const int rowmax = (qspan * a.s + (a.K - 1) * adj + 1) * a.P;
  g.nXrow = (rowmax + 63) / 64;
  const int want = (a.K * adj * a.P) % 32;
  g.XP = g.nXrow * 64 + want;  // pitch == K*dj*P (mod 32): the (c, k) columns of a B fragment hit distinct banks
  g.a_floats = BM * AP;
  g.buf_floats = g.a_floats + g.NCH * g.XP + BU;
  const size_t lds = 2ull * g.buf_floats * 4;
  if (lds > VCV_LDS_LIMIT) return -100;
  const long long U = (long long)a.Ta * a.P;
  g.nchunk_u = (int)((U + BU - 1) / BU);
  const long long total = (long long)a.B * g.nchunk_u;
  // split of the position chunks over grid.z: minimise (rounds of resident blocks) x (chunks per block + a
  // fixed prologue/atomic-epilogue cost of about two chunks)
  const long long tiles = (long long)g.nnt * g.nmt;
  const long long occ = lds * 2 <= VCV_LDS_LIMIT ? 2 : 1;
  const long long slots = 256 * occ;
  long long Z = 1;
  double best = 1e30;
  const size_t nw = (size_t)a.Mg * N;
  const long long zcap = a.slab ? (long long)(a.slab_floats / (int64_t)nw) : 1024;
  if (a.slab && zcap < 1) return VCV_EINVAL;
  for (long long z = 1; z <= total && z <= 1024 && z <= zcap; ++z) {
    const double rounds = (double)((tiles * z + slots - 1) / slots);
    const double cost = rounds * ((double)((total + z - 1) / z) + 2.0) / (double)occ;
    if (cost < best - 1e-9) best = cost, Z = z;
  }
  g.Z = (int)Z;
  const bool la = a.a_tf == VCV_TF_LEAKY, lb = a.b_tf == VCV_TF_LEAKY;
  // the bias-collecting variant exists for the transform-free `a` operand only (a Conv's dy)
  void (*kern)(const VcvWgradArgs, const WgGeom) =
      la ? (lb ? wgrad_dma_kernel<TM, TN, WM, WN, true, true, false> : wgrad_dma_kernel<TM, TN, WM, WN, true, false, false>)
         : (a.dbias ? (lb ? wgrad_dma_kernel<TM, TN, WM, WN, false, true, true> : wgrad_dma_kernel<TM, TN, WM, WN, false, false, true>)
                    : (lb ? wgrad_dma_kernel<TM, TN, WM, WN, false, true, false> : wgrad_dma_kernel<TM, TN, WM, WN, false, false, false>));
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VCV_EHIP;
  dim3 grid(g.nnt, g.nmt, g.Z), block(NT);
  const double flops = 2.0 * a.B * a.Mg * a.Cg * a.K * a.P * (double)a.Ta;
  const int tag[12] = {a.B, 1, a.Cg, a.Mg, a.K, a.Ta, a.P, a.s, g.Z, 2, BM * 1000 + BN, g.nXrow};
  hipEvent_t ev0, ev1;
  const double abytes = 4.0 * ((double)a.B * a.Mg * a.Ta * a.P + (double)a.B * a.Cg * a.Tb * a.P + (double)a.Mg * a.Cg * a.K);
  vcv_prof_events(VCV_PROF_WGRAD_DMA, flops, tag, 12, &ev0, &ev1, abytes);
  hipExtLaunchKernelGGL(kern, grid, block, (unsigned)lds, st, ev0, ev1, 0, a, g);
  if (a.slab)
    hipLaunchKernelGGL(wgrad_slab_finish_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, (const float*)a.slab, a.dw,
                       nw, g.Z, a.alpha);
  return vcv_check_launch();
}

}  // namespace

// returns -100 when the launch is not eligible for the DMA path (the caller falls back)
int vcv_wgrad_dma_try(const VcvWgradArgs& a, hipStream_t st) {
  const bool tf_ok = (a.a_tf == VCV_TF_NONE || a.a_tf == VCV_TF_LEAKY) && (a.b_tf == VCV_TF_NONE || a.b_tf == VCV_TF_LEAKY) &&
                     a.slope >= 0.f && a.slope < 1.f;
  const int N = a.Cg * a.K;
  const long long U = (long long)a.Ta * a.P;
  if (a.G != 1 || !tf_ok || a.transpose_out || a.Mg < 32 || N < 96 || U < 64 || a.s < 1) return -100;
  if (U * 4 >= (1ll << 31) || (long long)a.Tb * a.P * 4 >= (1ll << 31)) return -100;
  int rc = -100;
  if (a.Mg >= 128) {
    if (N >= 1024) rc = launch<2, 1, 2, 8>(a, st);  // 128x256, 16 waves
    if (rc == -100) rc = launch<2, 1, 2, 4>(a, st);  // 128x128, 8 waves: its LDS allows one workgroup per CU
    if (rc == -100) rc = launch<1, 1, 4, 2>(a, st);  // 128x64, 8 waves
    return rc;
  }
  if (a.Mg >= 64) {
    rc = launch<1, 1, 2, 4>(a, st);  // 64x128, 8 waves
    if (rc == -100) rc = launch<1, 1, 2, 2>(a, st);
    return rc;
  }
  return launch<1, 1, 1, 4>(a, st);
}

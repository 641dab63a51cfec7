// wgrad_dma.hip -- the MFMA weight-gradient kernel of the fp32 path (see conv_wgrad.hip for the GEMM mapping: M =
// channels of the un-shifted operand, N = (c, k) in the weight's memory order, reduction over positions, split over
// the grid and combined with fp32 atomics or per-split slabs).
//
// Both operands are activations, so nothing needs packing.  Per stage of 64 positions u = q*P + pc the LDS holds
//   As[m][u]     the un-shifted operand, row pitch 68 floats (rows past M and positions past the sequence end zero);
//   Xs[c][f]     the shifted operand's contiguous span of every channel of the column tile: all taps and all
//                positions of the stage read it at their offsets (floats outside the sequence are zero).
// Unit stride: tap k of position u sits at Xs[c][u + k*dj*P + const], so a lane reads FOUR consecutive positions of
// its row / column with one ds_read_b128 (the B side at 4-byte alignment, which the LDS serves at full rate --
// tools/scratch/lds_probe.hip) and feeds four v_mfma_f32_32x32x2_f32 per fragment pair: a quarter of the LDS
// instructions of a dword-per-MFMA loop.  Strided launches (the period discriminators' s = 3 convs, the generator's
// transposed convs) jump by (s-1)*P floats where the positions wrap to the next row: their B side reads dwords at a
// per-position offset table the producers write beside the stage (the A side still reads 16 bytes).
//
// Warp-specialised: NW MFMA waves + NP producer waves per workgroup.  The producers stage with 16-byte buffer loads
// into registers and 16-byte ds_writes; the loads of stage s+2 are issued as soon as stage s+1 is written, while
// stage s is multiplied.  What was measured on the way here (DiscP conv4 / conv3 weight gradients, TFLOP/s):
//   every wave issues `buffer_load_dword ... lds` DMAs at the top of its stage        83 / 59  (the issue phase -- ~12
//       cycles per DMA instruction -- and the MFMA phase of a stage add up: all waves leave the barrier together)
//   the same DMAs handed out between the MFMA groups                                  66 / 40
//   the DMAs issued by producer waves                                                 98 / 65  (a dword DMA holds
//       the LDS write port long enough to stall the fragment reads: with the X DMAs off 97, with 16-byte ds_writes
//       of the same bytes in their place 91)
//   register-staged producers that de-interleave strided rows by residue as they write                104 / 65
//       (~25 VALU instructions per element: one producer wave per SIMD cannot keep up)
// One barrier per stage; input leaky-ReLU (ResBlock / generator convs) is applied to the fragments as read.
#include "common.h"
#include "prof.h"
#include <cstring>
#include <unordered_map>

namespace {

constexpr int BU = 64, AP = BU + 4, MAXX = 12, TABN = BU + 32;

struct WgGeom {
  int NCH, CP, emin, G4, nmt, nnt, Z, nchunk_u, buf_floats, a_floats, tab_floats;
  float invG4, invP;
};

// The fragment reads are inline asm (the compiler splits a 4-byte-aligned 16-byte LDS load into dword pairs), so
// their completion is waited for by hand: `s_waitcnt lgkmcnt(0)` followed by an empty asm that redefines the
// fragment registers, which keeps every MFMA that consumes them behind the wait.
template <int OFF>
__device__ __forceinline__ f32x4 lds_rd128(unsigned addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
__device__ __forceinline__ float lds_rd32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pin(f32x4& f) { asm volatile("" : "+v"(f)); }
__device__ __forceinline__ void pin(float& f) { asm volatile("" : "+v"(f)); }
template <typename T, int N>
__device__ __forceinline__ void pin(T (&f)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) pin(f[i]);
}
// n / d and n % d for 0 <= n < 2^22 with inv = 1.0f / d (one correction step makes the float quotient exact)
__device__ __forceinline__ void divmod(int n, int d, float inv, int& q, int& r) {
  q = (int)((float)n * inv);
  r = n - q * d;
  if (r < 0) r += d, --q;
  if (r >= d) r -= d, ++q;
}
__device__ __forceinline__ f32x4 ld128(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
}

template <int TM, int TN, int WM, int WN, bool LA, bool LB, bool BIAS, bool STR>
__global__ void __launch_bounds__(64 * (WM * WN + (WM * WN >= 8 ? 4 : 2)))
wgrad_dma_kernel(const VcvWgradArgs p, const WgGeom tg) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN, NP = NW >= 8 ? 4 : 2;
  constexpr int MAXA = BM / (4 * NP);  // 16-byte loads per producer lane that cover the A tile
  extern __shared__ float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  const int nt = blockIdx.x, mt = blockIdx.y, z = blockIdx.z;
  const int K = p.K, Cg = p.Cg, Mg = p.Mg, P = p.P;
  const int N = Cg * K;
  const int n0 = nt * BN, m0 = mt * BM;
  const int cfirst = n0 / K;
  const int CP = tg.CP, NCH = tg.NCH;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
  const int U = p.Ta * P, TbP = p.Tb * P;
  const int total = p.B * tg.nchunk_u;
  if (z >= total) return;
  // first float of the X span of a stage inside its channel row (the image starts at this rounded down to a multiple
  // of four: a 16-byte group that begins before the buffer is zero as a whole, so none may straddle the row start)
  auto xspan0 = [&](int uc0) __attribute__((always_inline)) {
    return STR ? ((uc0 / P) * p.s + tg.emin) * P : uc0 + tg.emin * P;
  };

  if (wave >= NW) {
    // ------------------------------------------------------------------------------------------ producer waves
    const int pw = wave - NW;
    __builtin_amdgcn_s_setprio(3);  // few instructions, all of them on the critical path of the next stage
    f32x4 ra[MAXA], rx[MAXX];
    const int ngx = NCH * tg.G4;  // 16-byte groups of the X image of a stage
    // this lane's X groups: float offset inside the batch item's channel block, byte offset inside the LDS image
    // (groups past the image are loaded from offset 0 and written to a scratch slot behind the table)
    int xld[MAXX], xi4[MAXX];
    unsigned xst[MAXX];
#pragma unroll
    for (int t = 0; t < MAXX; ++t) {
      const int idx = (t * NP + pw) * 64 + lane;
      int cl, i4;
      divmod(idx, tg.G4, tg.invG4, cl, i4);
      xi4[t] = 4 * i4;
      xld[t] = idx < ngx ? cl * TbP + 4 * i4 : 0x20000000;
      xst[t] = (unsigned)(idx < ngx ? tg.a_floats + cl * CP + 4 * i4 : tg.a_floats + tg.tab_floats + NCH * CP) * 4u;
    }
    if (STR && pw == 1 && lane < TABN - BU) {  // the table's spare entries (read one group ahead, never used)
      asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + (unsigned)(tg.a_floats + NCH * CP + BU + lane) * 4u), "v"(0) : "memory");
      asm volatile("ds_write_b32 %0, %1" ::"v"(lds0 + (unsigned)(tg.buf_floats + tg.a_floats + NCH * CP + BU + lane) * 4u), "v"(0)
                   : "memory");
    }
    auto store = [&](int ch, int buf) __attribute__((always_inline)) {
      const int b = ch / tg.nchunk_u;
      const int uc0 = (ch - b * tg.nchunk_u) * BU;
      const unsigned base = lds0 + (unsigned)(buf * tg.buf_floats) * 4u;
      const bool tail = uc0 + BU > U;
#pragma unroll
      for (int t = 0; t < MAXA; ++t) {
        const int row = (t * NP + pw) * 4 + (lane >> 4), c4 = lane & 15;
        f32x4 v = ra[t];
        if (tail) {
          const int nv = U - uc0 - 4 * c4;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = e < nv ? v[e] : 0.f;
        }
        asm volatile("ds_write_b128 %0, %1" ::"v"(base + (unsigned)(row * AP + 4 * c4) * 4u), "v"(v) : "memory");
      }
      const int f0 = xspan0(uc0), f0a = f0 & ~3;
      if (STR && pw == 0) {
        // byte offset of position uc0 + lane inside a channel's image: its row advances s rows per q
        int q, pc;
        divmod(uc0 + lane, P, tg.invP, q, pc);
        const int off = ((q - uc0 / P) * p.s * P + pc + (f0 - f0a)) * 4;
        asm volatile("ds_write_b32 %0, %1" ::"v"(base + (unsigned)(tg.a_floats + NCH * CP + lane) * 4u), "v"(uc0 + lane < U ? off : 0)
                     : "memory");
      }
      const bool edge = f0a < 0 || f0a + 4 * tg.G4 > TbP;
#pragma unroll
      for (int t = 0; t < MAXX; ++t) {
        if ((t * NP + pw) * 64 < ngx) {
          f32x4 v = rx[t];
          if (edge) {
            const int f = f0a + xi4[t];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (unsigned)(f + e) < (unsigned)TbP ? v[e] : 0.f;
          }
          asm volatile("ds_write_b128 %0, %1" ::"v"(base + xst[t]), "v"(v) : "memory");
        }
      }
    };
    auto load = [&](int ch) __attribute__((always_inline)) {
      const int b = ch / tg.nchunk_u;
      const int uc0 = (ch - b * tg.nchunk_u) * BU;
      __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a + ((size_t)b * Mg + m0) * (size_t)U), 0,
                                                                    (int)((size_t)(Mg - m0) * U * 4), 0x00020000);
#pragma unroll
      for (int t = 0; t < MAXA; ++t) {
        const int row = (t * NP + pw) * 4 + (lane >> 4);
        ra[t] = ld128(rA, (unsigned)(row * U + uc0 + 4 * (lane & 15)) * 4u);
      }
      const int f0a = xspan0(uc0) & ~3;
      __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(p.b + ((size_t)b * Cg + cfirst) * (size_t)TbP), 0,
                                                                    (int)((size_t)(Cg - cfirst) * TbP * 4), 0x00020000);
#pragma unroll
      for (int t = 0; t < MAXX; ++t)
        if ((t * NP + pw) * 64 < ngx) {
          unsigned voff = (unsigned)(xld[t] + f0a) * 4u;
          asm volatile("" : "+v"(voff));  // keep it one register: see conv_pk.hip loadX (immediate-offset folding)
          rx[t] = ld128(rX, voff);
        }
    };
    load(z);
    store(z, 0);
    if (z + tg.Z < total) load(z + tg.Z);
    int bufi = 0;
    for (int ch = z; ch < total; ch += tg.Z) {
      lds_wait();
      __syncthreads();
      if (ch + tg.Z < total) {
        store(ch + tg.Z, bufi ^ 1);
        if (ch + 2 * tg.Z < total) load(ch + 2 * tg.Z);
      }
      bufi ^= 1;
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------- MFMA waves
  const int wm = wave / WN, wn = wave % WN;
  // byte offsets of this lane's fragment rows inside a stage buffer (the lane's half h reads positions 4h..4h+3 of
  // every group of eight)
  unsigned aofs[TM], nofs[TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) aofs[tm] = (unsigned)(((wm * TM + tm) * 32 + l31) * AP + 4 * h) * 4u;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int n = n0 + (wn * TN + tn) * 32 + l31;
    if (n > N - 1) n = N - 1;
    const int c = n / K, kw = n - c * K;
    const int e = kw * p.dj + p.off - tg.emin;  // >= 0: rows below the span's first row
    // unit stride: position i of the stage sits i floats further (plus the span's round-down, the same for every
    // stage: the stages start at multiples of 64); strided: the per-position table holds the rest
    nofs[tn] = (unsigned)(tg.a_floats + (c - cfirst) * CP + e * P + (STR ? 0 : ((tg.emin * P) & 3) + 4 * h)) * 4u;
  }
  const unsigned tabofs = (unsigned)(tg.a_floats + NCH * CP + 4 * h) * 4u;

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

  int bufi = 0;
  for (int ch = z; ch < total; ch += tg.Z) {
    __syncthreads();
    const unsigned base = lds0 + (unsigned)(bufi * tg.buf_floats) * 4u;
    const int uc0 = (ch % tg.nchunk_u) * BU;
    // a sequence's last stage multiplies only the sixteen-position groups that hold positions
    const int rem = U - uc0;
    const int npair = rem >= BU ? BU / 16 : (rem + 15) >> 4;
    unsigned aad[TM], bad[TN], tad = base + tabofs;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) aad[tm] = base + aofs[tm];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bad[tn] = base + nofs[tn];

    // (strided: the dword reads land in scalar registers b0s / b1s -- an asm output that is then moved into a vector
    // element would be copied before its data has arrived)
    f32x4 a0[TM], a1[TM], b0[TN], b1[TN], t0, t1;
    float b0s[TN][4], b1s[TN][4];
#define WG_LOAD(A, Bf, T, OFF)                                                                  \
  {                                                                                             \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) A[tm] = lds_rd128<OFF>(aad[tm]);          \
    if (!STR) {                                                                                 \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) Bf[tn] = lds_rd128<OFF>(bad[tn]);       \
    } else {                                                                                    \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) _Pragma("unroll") for (int e = 0; e < 4; ++e)  \
          Bf##s[tn][e] = lds_rd32(bad[tn] + __float_as_uint(T[e]));                             \
    }                                                                                           \
  }
#define WG_MMA(A, Bf)                                                                                              \
  {                                                                                                                \
    if (LA) {                                                                                                      \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) _Pragma("unroll") for (int e = 0; e < 4; ++e)              \
          A[tm][e] = fmaxf(A[tm][e], A[tm][e] * p.slope);                                                          \
    }                                                                                                              \
    if (LB) {                                                                                                      \
      _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) _Pragma("unroll") for (int e = 0; e < 4; ++e) {            \
        if (STR) Bf##s[tn][e] = fmaxf(Bf##s[tn][e], Bf##s[tn][e] * p.slope);                                       \
        else Bf[tn][e] = fmaxf(Bf[tn][e], Bf[tn][e] * p.slope);                                                    \
      }                                                                                                            \
    }                                                                                                              \
    if (BIAS && do_bias) {                                                                                         \
      _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) bsum[tm] += (A[tm][0] + A[tm][1]) + (A[tm][2] + A[tm][3]); \
    }                                                                                                              \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                                          \
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[tm][e], STR ? Bf##s[tn][e] : Bf[tn][e], acc[tm][tn], 0, 0, 0); \
  }
#define PIN_S(B) _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) _Pragma("unroll") for (int e = 0; e < 4; ++e) pin(B[tn][e]);
    // strided: the table entries of a group are read one group ahead of the fragments they address (the table has
    // 32 spare entries, so the reads past the stage stay inside it)
    if (STR) {
      t0 = lds_rd128<0>(tad);
      lds_wait();
      pin(t0);
      t1 = lds_rd128<32>(tad);
    }
    WG_LOAD(a0, b0, t0, 0)
    lds_wait();
    pin(a0);
    if (STR) { PIN_S(b0s) pin(t1); } else pin(b0);
    for (int pr = 0; pr < npair; ++pr) {
      // (the last pair's second load reads the eight floats after the stage: in the buffer, never multiplied)
      WG_LOAD(a1, b1, t1, 32)
      if (STR) t0 = lds_rd128<64>(tad);
      WG_MMA(a0, b0)
      lds_wait();
      pin(a1);
      if (STR) { PIN_S(b1s) pin(t0); } else pin(b1);
      WG_LOAD(a0, b0, t0, 64)
      if (STR) t1 = lds_rd128<96>(tad);
      WG_MMA(a1, b1)
      lds_wait();
      pin(a0);
      if (STR) { PIN_S(b0s) pin(t1); } else pin(b0);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) aad[tm] += 64;
      if (!STR) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bad[tn] += 64;
      }
      tad += 64;
    }
#undef PIN_S
#undef WG_MMA
#undef WG_LOAD
    bufi ^= 1;
  }

  if (do_bias) {
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
      const float s2 = bsum[tm] + __shfl_xor(bsum[tm], 32, 64);  // the two position halves of the lane halves
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

template <int TM, int TN, int WM, int WN, bool STR>
void (*pick_kernel(const VcvWgradArgs& a))(const VcvWgradArgs, const WgGeom) {
  const bool la = a.a_tf == VCV_TF_LEAKY, lb = a.b_tf == VCV_TF_LEAKY;
  // the bias-collecting variant exists for the transform-free `a` operand only (a Conv's dy)
  return la ? (lb ? wgrad_dma_kernel<TM, TN, WM, WN, true, true, false, STR> : wgrad_dma_kernel<TM, TN, WM, WN, true, false, false, STR>)
            : (a.dbias ? (lb ? wgrad_dma_kernel<TM, TN, WM, WN, false, true, true, STR> : wgrad_dma_kernel<TM, TN, WM, WN, false, false, true, STR>)
                       : (lb ? wgrad_dma_kernel<TM, TN, WM, WN, false, true, false, STR> : wgrad_dma_kernel<TM, TN, WM, WN, false, false, false, STR>));
}

struct WgPlan {
  WgGeom g;
  size_t lds;
  double cost;  // estimated microseconds; < 0: the tile does not fit
};

// Geometry of one tile shape for a launch, the split of the reduction over grid.z and an estimate of the time:
//   rounds of resident workgroups x (stages per workgroup x stage time x workgroups sharing the CU + a fixed
//   prologue) + the epilogue's Z x M x N atomic adds at the chip-wide atomic rate (MI355X_MICROARCH.md: ~1.3 TB/s)
//   [slab mode: the slab written and read back at streaming rates].
// The stage time prices the MFMAs at the rate the tile shape reaches in this kernel (LDS instructions per MFMA).
template <int TM, int TN, int WM, int WN, bool STR>
WgPlan plan(const VcvWgradArgs& a) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NP = WM * WN >= 8 ? 4 : 2, NT = 64 * (WM * WN + NP);
  WgPlan pl;
  pl.cost = -1.0;
  WgGeom& g = pl.g;
  const int N = a.Cg * a.K;
  g.nnt = vcv_cdiv(N, BN);
  g.nmt = vcv_cdiv(a.Mg, BM);
  g.NCH = (BN - 1) / a.K + 2;
  if (g.NCH > a.Cg + 1) g.NCH = a.Cg + 1;
  // tap row offsets k*dj + off span emin .. emin + emax rows
  const int jspan = (a.K - 1) * a.dj;
  g.emin = a.off + (jspan < 0 ? jspan : 0);
  const int emax = jspan < 0 ? -jspan : jspan;
  // floats of a channel row a stage reads: the rows of its positions (the first one partial in strided launches, whose
  // spans start at a row start) plus the taps' rows
  const int nq = STR ? (a.P - 1 + BU - 1) / a.P + 1 : 0;
  const int span = STR ? ((nq - 1) * a.s + emax + 1) * a.P : BU + emax * a.P;
  g.G4 = (span + 3 + 3) / 4;  // + up to three floats of round-down at the start
  g.invG4 = 1.0f / (float)g.G4, g.invP = 1.0f / (float)a.P;
  if ((long long)g.NCH * g.G4 > (long long)MAXX * NP * 64) return pl;
  g.a_floats = BM * AP;
  g.tab_floats = TABN;
  // channel pitch: the (c, k) columns of a B fragment start |dj|*P floats apart inside a channel; step the channels by
  // about K of those modulo the 64 banks -- when the LDS has room for the padding
  int want = (a.K * (emax / (a.K > 1 ? a.K - 1 : 1)) * a.P + 3) / 4 * 4 % 64;
  if (want < 8) want = 8;
  g.CP = 4 * g.G4;
  while (g.CP % 64 != want % 64) g.CP += 4;
  if (g.CP - 4 * g.G4 > 32) g.CP = 4 * g.G4 + 4;
  for (int attempt = 0; attempt < 2; ++attempt) {
    g.buf_floats = g.a_floats + g.NCH * g.CP + g.tab_floats + 16;
    pl.lds = 2ull * g.buf_floats * 4;
    if (pl.lds <= VCV_LDS_LIMIT) break;
    g.CP = 4 * g.G4 + ((4 * g.G4) % 32 == 0 ? 4 : 0);
  }
  if (pl.lds > VCV_LDS_LIMIT) return pl;
  const long long U = (long long)a.Ta * a.P;
  g.nchunk_u = (int)((U + BU - 1) / BU);
  const long long total = (long long)a.B * g.nchunk_u;
  // workgroups a CU holds (registers and LDS): asked once per kernel variant
  void (*kern)(const VcvWgradArgs, const WgGeom) = pick_kernel<TM, TN, WM, WN, STR>(a);
  static int occ_cache[2][2][2] = {};
  int& oc = occ_cache[a.a_tf == VCV_TF_LEAKY][a.b_tf == VCV_TF_LEAKY][a.dbias != nullptr];
  if (oc == 0) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, VCV_LDS_LIMIT);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, NT, 0) != hipSuccess || nb < 1) nb = 1;
    oc = nb;
  }
  long long occ = (long long)(VCV_LDS_LIMIT / pl.lds);
  if (occ > oc) occ = oc;
  if (occ < 1) occ = 1;
  // full-load MFMA rate of the tile shape relative to 128x256 (measured on the DiscP conv4 weight gradient: LDS
  // instructions per MFMA, MFMA waves per SIMD, producers per workgroup), 128x256 itself at 0.70 of the fp32 peak
  const double shape = BM == 128 ? (BN == 256 ? 1.0 : BN == 128 ? 0.93 : 0.77)
                     : BM == 64  ? (BN == 256 ? 0.905 : BN == 128 ? 0.76 : 0.58)
                                 : 0.62;
  const double rate = 0.70 * shape * (STR ? 0.93 : 1.0) * 157.3e12 / 256.0;  // per CU
  const double t_stage = 2.0 * BM * BN * BU / rate * 1e6, t_fix = 4.0;
  const long long tiles = (long long)g.nnt * g.nmt;
  const long long slots = 256 * occ;
  const size_t nw = (size_t)a.Mg * N;
  const long long zcap = a.slab ? (long long)(a.slab_floats / (int64_t)nw) : 1024;
  if (a.slab && zcap < 1) return pl;
  long long Z = 1;
  double best = 1e30;
  for (long long z = 1; z <= total && z <= 1024 && z <= zcap; ++z) {
    const double rounds = (double)((tiles * z + slots - 1) / slots);
    // workgroups sharing a CU: a partly filled CU runs each of them faster, but not in proportion
    const double r = tiles * z < slots ? (double)(tiles * z) / 256.0 : (double)occ;
    const double share = r <= 1.0 ? (0.3 + 0.7 / (double)occ) * (double)occ : (0.3 + 0.7 * r / (double)occ) * (double)occ;
    const double epi = a.slab ? (double)z * nw * 4.0 * (1.0 / 4e12 + 1.0 / 4e12) * 1e6 + 3.0 : (double)z * nw * 4.0 / 1.3e12 * 1e6;
    const double cost = rounds * ((double)((total + z - 1) / z) * t_stage * share + t_fix) + epi;
    if (cost < best - 1e-9) best = cost, Z = z;
  }
  g.Z = (int)Z;
  pl.cost = best;
  const bool verbose = vcv_tuning().wgrad_verbose != 0;
  if (verbose)
    fprintf(stderr, "wgrad plan M%d C%d K%d U%lld s%d tile %dx%d: occ %lld tiles %lld total %lld Z %lld cost %.1f us lds %zu\n", a.Mg, a.Cg, a.K,
            U, a.s, BM, BN, occ, tiles, total, Z, best, pl.lds);
  return pl;
}

template <int TM, int TN, int WM, int WN, bool STR>
int run(const VcvWgradArgs& a, const WgPlan& pl, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NP = WM * WN >= 8 ? 4 : 2, NT = 64 * (WM * WN + NP);
  const WgGeom& g = pl.g;
  void (*kern)(const VcvWgradArgs, const WgGeom) = pick_kernel<TM, TN, WM, WN, STR>(a);
  if (pl.lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds) != hipSuccess)
    return VCV_EHIP;
  const int N = a.Cg * a.K;
  const size_t nw = (size_t)a.Mg * N;
  dim3 grid(g.nnt, g.nmt, g.Z), block(NT);
  const double flops = 2.0 * a.B * a.Mg * a.Cg * a.K * a.P * (double)a.Ta;
  const int tag[12] = {a.B, 1, a.Cg, a.Mg, a.K, a.Ta, a.P, a.s, g.Z, 2, BM * 1000 + BN, g.G4};
  hipEvent_t ev0, ev1;
  const double abytes = 4.0 * ((double)a.B * a.Mg * a.Ta * a.P + (double)a.B * a.Cg * a.Tb * a.P + (double)a.Mg * a.Cg * a.K);
  vcv_prof_events(VCV_PROF_WGRAD_DMA, flops, tag, 12, &ev0, &ev1, abytes);
  VCV_LAUNCH_EV(kern, grid, block, (unsigned)pl.lds, st, ev0, ev1, a, g);
  if (a.slab)
    hipLaunchKernelGGL(wgrad_slab_finish_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, (const float*)a.slab, a.dw,
                       nw, g.Z, a.alpha);
  return vcv_check_launch();
}

// the tile shapes: <TM, TN, WM, WN> = 32x32 MFMA tiles per wave and waves per workgroup, rows x columns
#define WG_TILES(X) X(0, 2, 2, 2, 4) X(1, 2, 1, 2, 4) X(2, 1, 1, 4, 2) X(3, 1, 2, 2, 4) X(4, 1, 2, 2, 2) X(5, 1, 1, 2, 2) X(6, 1, 2, 1, 2)

// plans are memoised per launch shape: evaluating seven tile shapes x up to 1024 splits costs ~50 us of host time,
// which a launch-bound step (the 48 kHz full model) would pay two hundred times
struct PlanKey {
  int v[14];
  int64_t slab;
  bool operator==(const PlanKey& o) const { return memcmp(v, o.v, sizeof(v)) == 0 && slab == o.slab; }
};
struct PlanKeyHash {
  size_t operator()(const PlanKey& k) const {
    uint64_t h = 1469598103934665603ull ^ (uint64_t)k.slab;
    for (int i = 0; i < 14; ++i) h = (h ^ (uint64_t)(uint32_t)k.v[i]) * 1099511628211ull;
    return (size_t)h;
  }
};
struct PlanEntry { WgPlan pl; int which; };

template <bool STR>
int launch(const VcvWgradArgs& a, hipStream_t st) {
  const int only = vcv_tuning().wgrad_tile;  // tuning sweeps (the plan cache below is keyed on the shape only: set it before the first launch)
  static thread_local std::unordered_map<PlanKey, PlanEntry, PlanKeyHash> cache;
  const PlanKey key = {{a.B, a.Cg, a.Mg, a.Ta, a.Tb, a.P, a.K, a.s, a.dj, a.off, a.a_tf, a.b_tf, a.dbias != nullptr, a.slab != nullptr},
                       a.slab ? a.slab_floats : 0};
  auto hit = cache.find(key);
  WgPlan best;
  best.cost = -1.0;
  int which = -1;
  if (hit != cache.end()) best = hit->second.pl, which = hit->second.which;
  else {
#define WG_PLAN(I, TM, TN, WM, WN)                                                        \
  if ((only < 0 || only == I) && a.Mg >= 32 * TM * WM / 2 + 1 || (I == 6 && which < 0)) { \
    const WgPlan pl = plan<TM, TN, WM, WN, STR>(a);                                       \
    if (pl.cost >= 0.0 && (which < 0 || pl.cost < best.cost)) best = pl, which = I;       \
  }
  WG_TILES(WG_PLAN)
#undef WG_PLAN
    if (cache.size() > 4096) cache.clear();
    cache[key] = PlanEntry{best, which};
  }
  if (which < 0) return -100;
#define WG_RUN(I, TM, TN, WM, WN) \
  if (which == I) return run<TM, TN, WM, WN, STR>(a, best, st);
  WG_TILES(WG_RUN)
#undef WG_RUN
  return -100;
}

}  // namespace

static bool wgrad_dma_eligible(const VcvWgradArgs& a) {
  const bool tf_ok = (a.a_tf == VCV_TF_NONE || a.a_tf == VCV_TF_LEAKY) && (a.b_tf == VCV_TF_NONE || a.b_tf == VCV_TF_LEAKY) &&
                     a.slope >= 0.f && a.slope < 1.f;
  const int N = a.Cg * a.K;
  const long long U = (long long)a.Ta * a.P;
  if (a.G != 1 || !tf_ok || a.transpose_out || a.Mg < 32 || N < 96 || U < 64 || a.s < 1 || a.s > 8) return false;
  // 32-bit byte offsets inside one batch item of either operand
  if ((long long)a.Mg * U * 4 >= (1ll << 31) || (long long)a.Cg * a.Tb * a.P * 4 >= (1ll << 31)) return false;
  return true;
}

// returns -100 when the launch is not eligible for this kernel (the caller falls back)
int vcv_wgrad_dma_try(const VcvWgradArgs& a, hipStream_t st) {
  if (!wgrad_dma_eligible(a)) return -100;
  return a.s == 1 ? launch<false>(a, st) : launch<true>(a, st);
}

// 1: vcv_conv_wgrad hands this launch to the LDS-DMA kernel (wgrad_dma_kernel); 0: it runs on the register-staged
// conv_wgrad_kernel (tools/fallback_census.py lists those shapes)
extern "C" int vcv_conv_wgrad_takes_dma(const VcvWgradArgs* a) {
  return a && vcv_tuning().wgrad_dma != 0 && wgrad_dma_eligible(*a) ? 1 : 0;
}

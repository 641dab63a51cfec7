// wgrad_bf16.hip -- bf16-operand weight gradient (v_mfma_f32_32x32x16_bf16, fp32 accumulate) of the conv family:
//   dw[m, c, k] += alpha * sum_{b, u = (q, p)} tfa(a[b, m, u]) * tfb(x[b, c, xrow(u) + k * dj * P])
// (VcvWgradArgs of include/vcvits_hip.h; `a` = un-shifted operand, `x` = shifted operand, both fp32 activations).
//
// GEMM mapping: D[m][c] per tap, reduction over positions: one MFMA = 32 m x 32 c x 16 consecutive positions of ONE
// tap.  Both operands want "8 consecutive reduction elements per lane", i.e. 8 consecutive POSITIONS of one channel,
// while memory (and the staging loads) are position-contiguous per channel.  Both are therefore staged into
// channel-innermost bf16 LDS images (rows = positions, columns = channels) and read back COLUMN-major with
// ds_read_b64_tr_b16, the hardware transpose read: a tap is a ROW offset of the x image, so tap shifts, strides and
// the [row][P] layout of the period discriminators never touch alignment.
//   Ya [u (BU)][m (BM)]      row pitch BM*2 (+64 B pad)      Xb [x row (span)][c (BC)]    row pitch BC*2 (+64 B pad)
// Pitch = 64 (mod 256) bytes: the 4 rows of a transpose-read block (consecutive, or 3 apart for the stride-3 period
// convs) fall into different 64-byte quarters of the 256-byte bank row -> conflict-free, with plain linear addressing
// (per-tap offsets are added to a per-lane base; no swizzle arithmetic in the loop).
// Staging: a wave-instruction covers 16 positions x 32 channels (lane: position = lane >> 2, 8-channel group = lane & 3),
// so the 8 lanes of a ds_write_b128 group write 8 distinct 16-byte slots; global reads are 64-byte runs.
// Reduction split over grid.z; every block writes its partial tile to its own slab and wgrad_bf16_finish_kernel adds
// the slabs in a fixed order: deterministic, no atomics.
#include "common.h"
#include "prof.h"

extern "C" int vcv_conv_x3_get_terms(void);
extern "C" int vcv_conv_x3_get_all(void);

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

constexpr int BU64 = 64;  // positions per stage (the split-operand launches, whose images hold three planes, take 32)

struct WbGeom {
  int nmt, nct, ntg;   // m tiles, c tiles, tap groups
  int Z;               // reduction split (grid.z)
  int nchunk_u;        // stages per batch element
  int XR;              // staged x rows per stage (multiple of 16)
  int pa, pb;          // row pitches (bytes) of the two images
  int a_bytes, buf_bytes;
  int a_plane, x_plane;  // bytes between the term planes of an image (split-operand launches)
};

typedef __attribute__((address_space(3))) char lds_char;

__device__ __forceinline__ bf16x8 tr_pair(const lds_char* p0, const lds_char* p1) {
  // two transposed reads = the 8 reduction elements of one MFMA operand (rows 0-3 and 4-7 of the lane's half)
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)p1);
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// WM x WC waves tile the (m, c) block, WU waves split the 16-position steps of a stage; each wave owns TM x 1 MFMA
// tiles per tap and KT tap accumulators.
// NT = 1: operands rounded to bf16.  NT = 6 / 9: exact fp32 operands as three bf16 terms each (conv_x3.hip's arithmetic):
// three image planes per operand, NT MFMAs per fragment pair.
// NP = 0: every wave stages and multiplies.  NP > 0 (the split-operand launches): warp-specialised -- the NP waves after
// the MFMA waves do all the staging (loads of stage s + 2 in flight while stage s + 1 is converted and stage s is
// multiplied), as in wgrad_dma.hip / conv_x3.hip.
template <int WM, int WC, int WU, int KT, int MAXT, int NT, int NP = 0, int BU = BU64>
__global__ void __launch_bounds__(64 * (WM * WC * WU + NP))
wgrad_bf16_kernel(const VcvWgradArgs p, const WbGeom tg, float* __restrict__ slab) {
  constexpr int BM = 32 * WM, BC = 32 * WC, NW = WM * WC * WU;
  constexpr int NS = NP ? NP : NW;  // staging waves
  constexpr int PL = NT == 1 ? 1 : 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sw = NP ? wave - NW : wave;  // index among the staging waves (negative: an MFMA wave of a specialised launch)
  const int wu = wave / (WM * WC), wmc = wave % (WM * WC);
  const int wm = wmc / WC, wc = wmc % WC;
  const int h = lane >> 5;

  const int ct = blockIdx.x % tg.nct, tgi = blockIdx.x / tg.nct;
  const int mt = blockIdx.y, z = blockIdx.z;
  const int K = p.K, Cg = p.Cg, Mg = p.Mg, P = p.P;
  const int k0 = tgi * KT;
  const int kn = K - k0 < KT ? K - k0 : KT;  // taps of this block
  const int m0 = mt * BM, c0 = ct * BC;
  const int PA = tg.pa, PB = tg.pb;
  // x rows staged per stage: taps k0 .. k0+kn-1 of positions uc0 .. uc0+BU-1
  const int tap_lo = p.dj >= 0 ? k0 * p.dj : (k0 + kn - 1) * p.dj;  // smallest row offset of the block's taps

  const long long U = (long long)p.Ta * P;
  const long long TbP = (long long)p.Tb * P;
  const int total = p.B * tg.nchunk_u;

  f32x16 acc[KT];
#pragma unroll
  for (int k = 0; k < KT; ++k)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[k][e] = 0.f;

  // staging tasks: (32-channel quad, 64-position block); A tasks first, then X tasks.  Lane: 8-channel group lane & 3,
  // positions 4 * (lane >> 2) .. + 3 -> 8 x 16-byte loads (256-byte runs per 16 lanes), 4 x 16-byte LDS writes
  const int nqa = BM / 32, nqb = BC / 32;
  const int ntA = nqa, ntB = nqb * (tg.XR / 64);  // (an A task covers 64 positions; with BU = 32 its upper half idles)
  const int ntask = ntA + ntB;
  const int lpq = lane >> 2, lg8 = lane & 3;
  f32x4 xr[MAXT][8];

  auto load = [&](int ch) {
    const int b = ch / tg.nchunk_u;
    const int uc0 = (ch - b * tg.nchunk_u) * BU;
    const int qa = uc0 / P;
    // first staged x position, rounded DOWN to a multiple of 4 so that a lane's 4-position vector is never partly
    // negative (a negative offset puts the whole vector out of range); tab[] carries the remainder
    const int f0 = ((qa * p.s + p.off + tap_lo) * P) & ~3;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int task = sw + t * NS;
      if (task < ntask) {
        const bool isA = task < ntA;
        const int tt = isA ? task : task - ntA;
        const int nq = isA ? nqa : nqb;
        const int quad = tt % nq, pblk = tt / nq;
        const int chan0 = (isA ? m0 : c0) + quad * 32;
        const int climit = isA ? Mg : Cg;
        const long long rowlen = isA ? U : TbP;
        const float* base = isA ? p.a + (size_t)b * Mg * (size_t)U : p.b + (size_t)b * Cg * (size_t)TbP;
        // one descriptor per task over the quad's 32 channel rows (wave-uniform); the lane's channel row goes into its
        // offset.  Rows past the channel count fall outside the descriptor (-> 0); positions outside [0, rowlen) would
        // read the neighbouring row, so they are masked per element.
        const int pos0 = (isA ? uc0 : f0) + pblk * 64 + lpq * 4;
        const int rows = climit - chan0 < 32 ? climit - chan0 : 32;
        const unsigned rec = rows > 0 ? (unsigned)((long long)rows * rowlen * 4) : 0u;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(base + (size_t)chan0 * (size_t)rowlen), 0, (int)rec, 0x00020000);
        bool ok[4];
        const bool in_stage = !isA || lpq * 4 < BU;  // (BU = 32: positions 32 .. 63 of an A task belong to the next stage)
#pragma unroll
        for (int j = 0; j < 4; ++j) ok[j] = in_stage && pos0 + j >= 0 && pos0 + j < rowlen;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned voff = (unsigned)(((long long)(lg8 * 8 + e) * rowlen + pos0) * 4);
          f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = ok[j] ? v[j] : 0.f;
          xr[t][e] = v;
        }
      }
    }
  };
  // bias gradient (p.dbias): row sums of `a`, taken from the fp32 staging registers of the A tasks by the blocks of the
  // first (channel tile, tap group); an A task = one 32-channel quad, always staged by the same wave (t == 0)
  const bool do_bias = p.dbias != nullptr && blockIdx.x == 0 && sw >= 0 && sw < ntA;
  float bsum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
  auto store = [&](int buf) {
    char* Ya = smem + buf * tg.buf_bytes;
    char* Xb = Ya + tg.a_bytes;
    if (do_bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bsum[e] += (xr[0][e][0] + xr[0][e][1]) + (xr[0][e][2] + xr[0][e][3]);
    }
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int task = sw + t * NS;
      if (task < ntask) {
        const bool isA = task < ntA;
        const int tt = isA ? task : task - ntA;
        const int nq = isA ? nqa : nqb;
        const int quad = tt % nq, pblk = tt / nq;
        const bool lk = isA ? p.a_tf == VCV_TF_LEAKY : p.b_tf == VCV_TF_LEAKY;
        char* img = isA ? Ya : Xb;
        const int pitch = isA ? PA : PB;
        const int plane = isA ? tg.a_plane : tg.x_plane;  // bytes between the term planes of an image
        if (isA && lpq * 4 >= BU) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bf16x8 v, v1, v2;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = xr[t][e][j];
            if (lk) f = fmaxf(f, f * p.slope);
            v[e] = (__bf16)f;
            if (PL == 3) {
              const float r1 = f - (float)v[e];  // exact
              v1[e] = (__bf16)r1;
              v2[e] = (__bf16)(r1 - (float)v1[e]);
            }
          }
          char* dst = img + (size_t)(pblk * 64 + lpq * 4 + j) * pitch + (quad * 4 + lg8) * 16;
          *reinterpret_cast<bf16x8*>(dst) = v;
          if (PL == 3) {
            *reinterpret_cast<bf16x8*>(dst + plane) = v1;
            *reinterpret_cast<bf16x8*>(dst + 2 * plane) = v2;
          }
        }
      }
    }
  };
  // tab[u] = x row (relative to the staged span) that tap offset tap_lo of position u reads
  auto build_tab = [&](int ch, int buf) {
    int* tab = (int*)(smem + buf * tg.buf_bytes + tg.a_bytes + PL * tg.x_plane);
    const int tid = (int)threadIdx.x - (NP ? NW * 64 : 0);  // the first staging wave builds it
    if (tid >= 0 && tid < BU) {
      const int b = ch / tg.nchunk_u;
      const int uc0 = (ch - b * tg.nchunk_u) * BU;
      const int qa = uc0 / P;
      const long long u = (long long)uc0 + tid;
      const int fs = (qa * p.s + p.off + tap_lo) * P;
      int t = fs - (fs & ~3);  // the span starts at a multiple of 4 (see load)
      if (u < U) {
        const int q = (int)(u / P), pc = (int)(u - (long long)q * P);
        t += (q - qa) * p.s * P + pc;
      }
      tab[tid] = t * PB;
    }
  };

  // per-lane constants of the transposed reads: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3
  const int li = lane & 15, lq = li >> 2, lp = li & 3, g1 = (lane >> 4) & 1;
  const int colA = ((wm * 32 + 16 * g1 + 4 * lp) * 2);
  const int colB = ((wc * 32 + 16 * g1 + 4 * lp) * 2);
  const int rowl = 8 * h + lq;  // the lane's row inside a 16-position step (second read: + 4)

  if (NP && wave >= NW) {
    // ---- producer waves: stage ch + 1 is converted and written while stage ch is multiplied; the loads of stage
    // ch + 2 are issued right after and have that whole stage to arrive
    if (z < total) {
      __builtin_amdgcn_s_setprio(3);
      load(z);
      store(0);
      build_tab(z, 0);
      if (z + tg.Z < total) load(z + tg.Z);
      int bufi = 0;
      for (int ch = z; ch < total; ch += tg.Z) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ch + tg.Z < total) {
          store(bufi ^ 1);
          build_tab(ch + tg.Z, bufi ^ 1);
          if (ch + 2 * tg.Z < total) load(ch + 2 * tg.Z);
        }
        bufi ^= 1;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else if (z < total) {
    if (!NP) {
      load(z);
      store(0);
      build_tab(z, 0);
    }
    __syncthreads();
    int bufi = 0;
    for (int ch = z; ch < total; ch += tg.Z) {
      const bool more = !NP && ch + tg.Z < total;
      if (more) load(ch + tg.Z);
      const lds_char* Ya = (const lds_char*)smem + bufi * tg.buf_bytes;  // 32-bit LDS addresses from here on
      const lds_char* Xb = Ya + tg.a_bytes;
      const int* tab = (const int*)(smem + bufi * tg.buf_bytes + tg.a_bytes + PL * tg.x_plane);
      for (int i16 = wu; i16 < BU / 16; i16 += WU) {
        const int ur = i16 * 16 + rowl;
        bf16x8 a[PL];
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) a[pl] = tr_pair(Ya + pl * tg.a_plane + ur * PA + colA, Ya + pl * tg.a_plane + (ur + 4) * PA + colA);
        const lds_char* x0 = Xb + tab[ur] + colB;
        const lds_char* x1 = Xb + tab[ur + 4] + colB;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          if (k < kn) {
            const int ro = ((k0 + k) * p.dj - tap_lo) * P * PB;
            bf16x8 bb[PL];
#pragma unroll
            for (int pl = 0; pl < PL; ++pl) bb[pl] = tr_pair(x0 + pl * tg.x_plane + ro, x1 + pl * tg.x_plane + ro);
            if (PL == 1) {
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[0], acc[k], 0, 0, 0);
            } else {  // small terms first (literal plane indices: see conv_x3.hip)
              if (NT == 9) {
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1], bb[PL - 1], acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL / 2], bb[PL - 1], acc[k], 0, 0, 0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1], bb[PL / 2], acc[k], 0, 0, 0);
              }
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[PL - 1], acc[k], 0, 0, 0);
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL - 1], bb[0], acc[k], 0, 0, 0);
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL / 2], bb[PL / 2], acc[k], 0, 0, 0);
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[PL / 2], acc[k], 0, 0, 0);
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PL / 2], bb[0], acc[k], 0, 0, 0);
              acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], bb[0], acc[k], 0, 0, 0);
            }
          }
        }
      }
      if (more) {
        store(bufi ^ 1);
        build_tab(ch + tg.Z, bufi ^ 1);
      }
      __syncthreads();
      bufi ^= 1;
    }
  }

  if (do_bias) {
    // lanes with the same lane & 3 hold the same 8 channels (different positions): reduce over lane bits 2..5
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = bsum[e];
      v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
      const int ml = m0 + sw * 32 + lg8 * 8 + e;
      if (lpq == 0 && ml < Mg) unsafeAtomicAdd(p.dbias + ml, v);
    }
  }
  if (NP && wave >= NW) return;

  // cross-wave reduction over the WU position-split waves (through LDS, one round per extra wave)
  if (WU > 1) {
    float* red = (float*)smem;  // [WM*WC][KT][16][64]
    for (int r = 1; r < WU; ++r) {
      __syncthreads();
      if (wu == r) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int e = 0; e < 16; ++e) red[((wmc * KT + k) * 16 + e) * 64 + lane] = acc[k][e];
      }
      __syncthreads();
      if (wu == 0) {
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[k][e] += red[((wmc * KT + k) * 16 + e) * 64 + lane];
      }
    }
    if (wu != 0) return;
  }

  // partial tile -> slab z, laid out [k][m][c]: the 32 lanes of an accumulator register hold 32 consecutive c of one
  // (k, m) row, so every store instruction writes two 128-byte runs (the dw order [m][c][k] would scatter 64 lanes over
  // 64 cache lines); wgrad_bf16_finish_kernel transposes back through LDS
  float* out = slab + (size_t)z * ((size_t)Mg * Cg * K);
  const int c = c0 + wc * 32 + (lane & 31);
  if (c < Cg) {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      if (k < kn) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ml = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (ml < Mg) out[((size_t)(k0 + k) * Mg + ml) * Cg + c] = acc[k][e];
        }
      }
    }
  }
}

// dw[m][c][k] += alpha * sum_z slab[z][k][m][c], slabs added in a fixed order (deterministic).  A workgroup is 8 groups
// of 32 lanes over a block of 32 c (slab reads in 128-byte runs, the 32 x K block of dw written as one contiguous run
// after a transpose through LDS).  Many slabs (ROWS == false): the 8 groups are z-lanes of ONE m row and meet in LDS.
// Few slabs (ROWS == true): each group owns its own m row and walks all Z slabs.
template <bool ROWS>
__global__ void __launch_bounds__(256) wgrad_bf16_finish_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                int M, int C, int K, int Z, float alpha) {
  __shared__ float red[8][16][33];
  const int cb = blockIdx.x * 32;
  const int cl = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int m = ROWS ? blockIdx.y * 8 + zl : blockIdx.y;
  const size_t n = (size_t)M * C * K;
  const bool ok = cb + cl < C && m < M;
  for (int k = 0; k < K; ++k) {
    const size_t off = ((size_t)k * M + m) * C + cb + cl;
    float a0 = 0.f, a1 = 0.f;
    if (ok) {
      const int step = ROWS ? 1 : 8;
      int z = ROWS ? 0 : zl;
      for (; z + step < Z; z += 2 * step) {
        a0 += slab[(size_t)z * n + off];
        a1 += slab[(size_t)(z + step) * n + off];
      }
      if (z < Z) a0 += slab[(size_t)z * n + off];
    }
    red[zl][k][cl] = a0 + a1;
  }
  __syncthreads();
  if (ROWS) {
    if (m < M)
      for (int idx = cl; idx < 32 * K; idx += 32) {
        const int c = idx / K, k = idx - c * K;
        if (cb + c < C) dw[((size_t)m * C + cb + c) * K + k] += alpha * red[zl][k][c];
      }
  } else {
    for (int idx = threadIdx.x; idx < 32 * K; idx += 256) {
      const int c = idx / K, k = idx - c * K;
      if (cb + c < C) {
        float sum = red[0][k][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) sum += red[q][k][c];
        dw[((size_t)m * C + cb + c) * K + k] += alpha * sum;
      }
    }
  }
}

// The same reduction with 16-byte slab reads and the slabs' loads of a thread all in flight at once (round 5: the kernel
// above kept two 4-byte loads per thread outstanding and ran at ~2 TB/s -- 4-6 % of the bf16-mode steps).  A workgroup is
// 32 groups of 8 lanes; a group reads the 32 c of one (k, m) row of a slab as eight float4.  The 32 groups are ZG z-lanes
// x MR = 32 / ZG rows of m; a group walks the slabs z = zg, zg + ZG, ... (ZU of them per round) for all K taps, so a
// thread has KB x ZU independent 16-byte loads outstanding.  z-lanes meet by xor-shuffles inside a wave (fixed tree), waves
// through LDS, slabs in a fixed order: deterministic.  KB >= K (register accumulators); needs C % 4 == 0.
template <int ZG, int KB>
__global__ void __launch_bounds__(256) wgrad_bf16_finish4_kernel(const float* __restrict__ slab, float* __restrict__ dw, int M,
                                                                 int C, int K, int Z, float alpha) {
  constexpr int MR = 32 / ZG;
  constexpr int ZU = KB >= 11 ? 1 : 16 / KB;
  constexpr int P = ZG > 8 ? ZG / 8 : 1;  // partial sums per m row left after the in-wave tree
  constexpr int ZW = ZG < 8 ? ZG : 8;     // z-lanes of one m row inside a wave
  __shared__ float red[P * MR][KB][33];
  const int t = threadIdx.x, l8 = t & 7, grp = t >> 3;
  const int zg = grp % ZG, mr = grp / ZG;
  const int m = blockIdx.y * MR + mr, cb = blockIdx.x * 32, c = cb + l8 * 4;
  const size_t n = (size_t)M * C * K, kstride = (size_t)M * C;
  float4 acc[KB];
#pragma unroll
  for (int k = 0; k < KB; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (m < M && c < C) {
    const float* base = slab + (size_t)m * C + c;
    for (int z0 = zg; z0 < Z; z0 += ZG * ZU) {
      float4 v[ZU][KB];
#pragma unroll
      for (int u = 0; u < ZU; ++u) {
        const int z = z0 + u * ZG;
#pragma unroll
        for (int k = 0; k < KB; ++k)
          v[u][k] = (z < Z && k < K) ? *(const float4*)(base + (size_t)z * n + (size_t)k * kstride) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < ZU; ++u)
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          acc[k].x += v[u][k].x; acc[k].y += v[u][k].y; acc[k].z += v[u][k].z; acc[k].w += v[u][k].w;
        }
    }
  }
#pragma unroll
  for (int k = 0; k < KB; ++k) {
    if (k < K) {
#pragma unroll
      for (int o = 8; o < 8 * ZW; o <<= 1) {
        acc[k].x += __shfl_xor(acc[k].x, o, 64); acc[k].y += __shfl_xor(acc[k].y, o, 64);
        acc[k].z += __shfl_xor(acc[k].z, o, 64); acc[k].w += __shfl_xor(acc[k].w, o, 64);
      }
      if ((zg & (ZW - 1)) == 0) {
        float* r = &red[mr * P + (ZG > 8 ? zg >> 3 : 0)][k][l8 * 4];
        r[0] = acc[k].x; r[1] = acc[k].y; r[2] = acc[k].z; r[3] = acc[k].w;
      }
    }
  }
  __syncthreads();
  const int per = 32 * K;
  for (int idx = t; idx < MR * per; idx += 256) {
    const int r = idx / per, q = idx - r * per;
    const int cc = q / K, k = q - cc * K;
    const int mm = blockIdx.y * MR + r;
    if (mm < M && cb + cc < C) {
      float sum = red[r * P][k][cc];
#pragma unroll
      for (int p = 1; p < P; ++p) sum += red[r * P + p][k][cc];
      dw[((size_t)mm * C + cb + cc) * K + k] += alpha * sum;
    }
  }
}

template <int ZG>
void launch_finish4(const float* slab, float* dw, int M, int C, int K, int Z, float alpha, hipStream_t st) {
  const dim3 grid((unsigned)vcv_cdiv(C, 32), (unsigned)vcv_cdiv(M, 32 / ZG)), block(256);
#define WB_F4(kb) hipLaunchKernelGGL((wgrad_bf16_finish4_kernel<ZG, kb>), grid, block, 0, st, slab, dw, M, C, K, Z, alpha)
  if (K <= 1) WB_F4(1);
  else if (K <= 3) WB_F4(3);
  else if (K <= 5) WB_F4(5);
  else if (K <= 8) WB_F4(8);
  else if (K <= 11) WB_F4(11);
  else WB_F4(16);
#undef WB_F4
}

int g_force_cand = -1, g_force_z = -1;  // tuning probe (tools/wgrad_variant_sweep.py): -1 = the library's choice

constexpr int MAXT = 2;
constexpr int MAXT_WS = 3;  // staging tasks per producer wave of a split-operand launch

struct Cfg { int WM, WC, WU, KT; };

bool geometry(const VcvWgradArgs& a, const Cfg& c, WbGeom& g, size_t& lds, int PL = 1) {
  const int BU = PL == 1 ? BU64 : 32;
  // producer waves for the bf16 launches too (round 3: 163 -> 176 TFLOP/s in the bf16 step); tuning key wgrad_bf16_ws = 0 keeps
  // every wave staging
  const bool ws1 = vcv_tuning().wgrad_bf16_ws != 0;
  const bool ws = PL == 3 || ws1;
  const int NS = ws ? 4 : c.WM * c.WC * c.WU, maxt = ws ? MAXT_WS : MAXT;  // staging waves, tasks each
  const int BM = 32 * c.WM, BC = 32 * c.WC, NW = c.WM * c.WC * c.WU;
  g.nmt = vcv_cdiv(a.Mg, BM);
  g.nct = vcv_cdiv(a.Cg, BC);
  g.ntg = vcv_cdiv(a.K, c.KT);
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  const int qspan = (BU - 1) / a.P + 1;
  const int kspan = (c.KT < a.K ? c.KT : a.K) - 1;
  const int rowmax = (qspan * a.s + kspan * adj + 1) * a.P;
  g.XR = (rowmax + 3 + 63) & ~63;  // + 3: the span start is rounded down to a multiple of 4
  auto pitch = [](int w) { const int b = w * 2; return b >= 128 ? b + 64 : b; };
  g.pa = pitch(BM);
  g.pb = pitch(BC);
  g.a_plane = BU * g.pa;
  g.x_plane = g.XR * g.pb;
  g.a_bytes = PL * g.a_plane;
  g.buf_bytes = g.a_bytes + PL * g.x_plane + BU * 4;
  lds = 2ull * g.buf_bytes;
  const size_t red = c.WU > 1 ? (size_t)c.WM * c.WC * c.KT * 16 * 64 * 4 : 0;
  if (red > lds) lds = red;
  if (lds > VCV_LDS_LIMIT) return false;
  const int ntask = (BM / 32) + (BC / 32) * (g.XR / 64);
  if (ntask > maxt * NS) return false;
  if (PL == 3 && BU / 16 < c.WU) return false;  // (32-position stages: every position-split wave needs a 16-position step)
  const long long Uu = (long long)a.Ta * a.P;
  g.nchunk_u = (int)((Uu + BU - 1) / BU);
  return true;
}

template <int WM, int WC, int WU, int KT, int NT>
int launch(const VcvWgradArgs& a, const WbGeom& g0, size_t lds, float* scratch, int64_t scratch_floats, hipStream_t st) {
  WbGeom g = g0;
  const long long total = (long long)a.B * g.nchunk_u;
  const long long tiles = (long long)g.nmt * g.nct * g.ntg;
  const long long occ = lds * 2 <= VCV_LDS_LIMIT ? 2 : 1;
  const long long slots = 256 * occ;
  const size_t n = (size_t)a.Mg * a.Cg * a.K;
  long long Z = 1;
  double best = 1e30;
  for (long long z = 1; z <= total && z <= 512; ++z) {
    if ((size_t)z * n > (size_t)scratch_floats) break;
    const double rounds = (double)((tiles * z + slots - 1) / slots);
    const double cost = rounds * ((double)((total + z - 1) / z) + 2.0) / (double)occ + 0.02 * z;  // + slab traffic
    if (cost < best - 1e-9) best = cost, Z = z;
  }
  if (g_force_z > 0) {
    Z = g_force_z;
    while (Z > 1 && ((size_t)Z * n > (size_t)scratch_floats || Z > total)) --Z;
  }
  g.Z = (int)Z;
  void (*kern)(const VcvWgradArgs, const WbGeom, float*);
  const bool ws1 = vcv_tuning().wgrad_bf16_ws != 0;
  if constexpr (NT == 1) kern = ws1 ? wgrad_bf16_kernel<WM, WC, WU, KT, MAXT_WS, NT, 4, BU64> : wgrad_bf16_kernel<WM, WC, WU, KT, MAXT, NT>;
  else kern = wgrad_bf16_kernel<WM, WC, WU, KT, MAXT_WS, NT, 4, 32>;  // 4 producer waves, 32-position stages
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VCV_EHIP;
  dim3 grid(g.nct * g.ntg, g.nmt, g.Z), block(64 * (WM * WC * WU + ((NT == 1 && !ws1) ? 0 : 4)));
  const double flops = 2.0 * a.B * a.Mg * a.Cg * a.K * a.P * (double)a.Ta;
  const int tag[12] = {a.B, NT == 1 ? 2 : 3, a.Cg, a.Mg, a.K, a.Ta, a.P, a.s, g.Z, 2, WM * 32 * 1000 + WC * 32, NT * 100 + KT};
  const double abytes = 4.0 * ((double)a.B * a.Mg * a.Ta * a.P + (double)a.B * a.Cg * a.Tb * a.P + (double)a.Mg * a.Cg * a.K);
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_WGRAD_DMA, flops, tag, 12, &ev0, &ev1, abytes, NT * flops / VCV_PEAK_BF16_MFMA);
  VCV_LAUNCH_EV(kern, grid, block, (unsigned)lds, st, ev0, ev1, a, g, scratch);
  const bool finish_scalar = !vcv_tuning().wgrad_finish_vec;  // (A/B switch: the 4-byte kernel)
  if (!finish_scalar && (a.Cg & 3) == 0 && ((uintptr_t)scratch & 15) == 0) {
    if (g.Z <= 4) launch_finish4<4>(scratch, a.dw, a.Mg, a.Cg, a.K, g.Z, a.alpha, st);
    else if (g.Z <= 8) launch_finish4<8>(scratch, a.dw, a.Mg, a.Cg, a.K, g.Z, a.alpha, st);
    else if (g.Z <= 16) launch_finish4<16>(scratch, a.dw, a.Mg, a.Cg, a.K, g.Z, a.alpha, st);
    else launch_finish4<32>(scratch, a.dw, a.Mg, a.Cg, a.K, g.Z, a.alpha, st);
  } else if (g.Z <= 12)
    hipLaunchKernelGGL(wgrad_bf16_finish_kernel<true>, dim3((unsigned)vcv_cdiv(a.Cg, 32), (unsigned)vcv_cdiv(a.Mg, 8)), dim3(256), 0,
                       st, (const float*)scratch, a.dw, a.Mg, a.Cg, a.K, g.Z, a.alpha);
  else
    hipLaunchKernelGGL(wgrad_bf16_finish_kernel<false>, dim3((unsigned)vcv_cdiv(a.Cg, 32), (unsigned)a.Mg), dim3(256), 0, st,
                       (const float*)scratch, a.dw, a.Mg, a.Cg, a.K, g.Z, a.alpha);
  return vcv_check_launch();
}

bool pick(const VcvWgradArgs& a, Cfg& c, WbGeom& g, size_t& lds, int PL = 1) {
  const bool tf_ok = (a.a_tf == VCV_TF_NONE || a.a_tf == VCV_TF_LEAKY) && (a.b_tf == VCV_TF_NONE || a.b_tf == VCV_TF_LEAKY) &&
                     a.slope >= 0.f && a.slope < 1.f;
  const long long U = (long long)a.Ta * a.P;
  if (a.G != 1 || !tf_ok || a.transpose_out || a.Mg < 32 || a.Cg < 16 || a.K > 16 || U * a.B < 256 || a.s < 1 || a.s > 3)
    return false;
  if (U * 4 >= (1ll << 31) || (long long)a.Tb * a.P * 4 >= (1ll << 31)) return false;
  c.KT = a.K == 1 ? 1 : a.K <= 3 ? 3 : a.K <= 5 ? 5 : (a.K == 7 || a.K == 8 || a.K >= 15) ? 8 : 6;
  // (three term planes: eight tap accumulators + nine fragments do not fit 256 registers; K = 7 runs as 4 + 3 taps)
  if (PL == 3 && c.KT == 8) c.KT = 4;
  // (WM, WC, WU) candidates, widest tile first; a candidate that does not fit the LDS / staging budget (long x spans of
  // the wide-period layouts) falls through to a narrower channel tile
  static const int cand[6][3] = {{4, 2, 1}, {4, 1, 2}, {2, 2, 2}, {2, 1, 4}, {1, 2, 4}, {1, 1, 8}};
  for (int i = 0; i < 6; ++i) {
    if (g_force_cand >= 0 && i != g_force_cand) continue;
    const int bm = 32 * cand[i][0], bc = 32 * cand[i][1];
    if (bm > 32 && bm > a.Mg) continue;
    if (bc > 32 && bc > ((a.Cg + 31) & ~31)) continue;
    c.WM = cand[i][0]; c.WC = cand[i][1]; c.WU = cand[i][2];
    if (geometry(a, c, g, lds, PL)) {
      // split-operand launches: three planes per image leave the wide-period layers (long x spans) only 32-channel
      // tiles, and K = 7 / 8 run as two tap groups: measured in the step (tools/prof_compare.py) the kernel is ahead of
      // the fp32 one (wgrad_dma.hip) with 64-channel tiles and K = 5 or K >= 9 -- 161-168 vs 105-108 TFLOP/s on the
      // 1024-channel period layers, 118-133 vs 77-97 on the generator's k = 11 layers -- and behind or level elsewhere
      if (PL == 3 && !vcv_conv_x3_get_all() && !(c.WC == 2 && (a.K == 5 || a.K >= 9))) return false;
      return true;
    }
  }
  return false;
}


int64_t scratch_want(const VcvWgradArgs* a, int PL) {
  Cfg c;
  WbGeom g;
  size_t lds;
  if (!a || !pick(*a, c, g, lds, PL)) return 0;
  const long long total = (long long)a->B * g.nchunk_u;
  const long long tiles = (long long)g.nmt * g.nct * g.ntg;
  long long z = (512 + tiles - 1) / tiles;
  if (z > total) z = total;
  if (z < 1) z = 1;
  const long long n = (long long)a->Mg * a->Cg * a->K;
  while (z > 1 && z * n > (64ll << 20)) --z;  // at most 256 MB of slabs
  if (z > 512) z = 512;
  return (int64_t)(z * n);
}

#define WB_CASE(wm, wc, wu, kt) \
  if (c.WM == wm && c.WC == wc && c.WU == wu && c.KT == kt) return launch<wm, wc, wu, kt, NT>(*a, g, lds, scratch, scratch_floats, st)
#define WB_KT(wm, wc, wu)                                                                        \
  WB_CASE(wm, wc, wu, 1); WB_CASE(wm, wc, wu, 3); WB_CASE(wm, wc, wu, 5); WB_CASE(wm, wc, wu, 6); \
  if constexpr (NT == 1) { WB_CASE(wm, wc, wu, 8); } else { WB_CASE(wm, wc, wu, 4); }

template <int NT>
int run(const VcvWgradArgs* a, float* scratch, int64_t scratch_floats, void* stream) {
  Cfg c;
  WbGeom g;
  size_t lds;
  if (!a || !scratch || !pick(*a, c, g, lds, NT == 1 ? 1 : 3)) return VCV_EINVAL;
  if (scratch_floats < (int64_t)a->Mg * a->Cg * a->K) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  WB_KT(4, 2, 1);
  WB_KT(4, 1, 2);
  WB_KT(2, 2, 2);
  WB_KT(2, 1, 4);
  WB_KT(1, 2, 4);
  WB_KT(1, 1, 8);
  return VCV_EINVAL;
}

}  // namespace

// Tuning probe: fix the tile candidate (0..5: 128x64, 128x32 x2, 64x64 x2, 64x32 x4, 32x64 x4, 32x32 x8; -1 = first that fits)
// and the reduction split Z (> 0; -1 = the cost model) of every launch.
extern "C" int vcv_wgrad_bf16_set_force(int cand, int z) {
  g_force_cand = cand;
  g_force_z = z;
  return VCV_OK;
}

// Scratch floats the launch wants (0: not eligible -> the caller uses vcv_conv_wgrad).  The kernel takes any scratch
// >= Mg*Cg*K floats and splits the reduction as far as the scratch allows.
extern "C" int64_t vcv_wgrad_bf16_scratch(const VcvWgradArgs* a) { return scratch_want(a, 1); }
extern "C" int vcv_wgrad_bf16(const VcvWgradArgs* a, float* scratch, int64_t scratch_floats, void* stream) {
  return run<1>(a, scratch, scratch_floats, stream);
}

// The same kernel on exact fp32 operands split into three bf16 terms each (the arithmetic of conv_x3.hip; the number of
// product terms is the one set with vcv_conv_x3_set_terms): fp32 weight gradients at the bf16 MFMA rate.
extern "C" int64_t vcv_wgrad_x3_scratch(const VcvWgradArgs* a) { return scratch_want(a, 3); }
extern "C" int vcv_wgrad_x3(const VcvWgradArgs* a, float* scratch, int64_t scratch_floats, void* stream) {
  return vcv_conv_x3_get_terms() == 6 ? run<6>(a, scratch, scratch_floats, stream) : run<9>(a, scratch, scratch_floats, stream);
}

// conv_x3.hip -- fp32 convolutions on the bf16 matrix pipe by EXACT operand splitting (vcv_conv_x3_*): the same
// forward-type / phased launch family as conv_pk.hip, same epilogue, same results to fp32 rounding.
//
// Why: v_mfma_f32_32x32x2_f32 peaks at 157 TFLOP/s; v_mfma_f32_32x32x16_bf16 at 2.5 PFLOP/s -- sixteen times the
// multiply-accumulates per cycle.  An fp32 number is the exact sum of three bf16 numbers,
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)
// (8 + 8 + 8 significand bits; both subtractions are exact in fp32, and bf16 has fp32's exponent range), so an fp32
// product is the sum of nine bf16 products, each of which the matrix core forms exactly and adds in fp32:
//     a * b = sum_{i, j} a_i * b_j .
// NTERM = 9 evaluates all of them: the operands enter with all 24 bits, every partial product is exact, and the fp32
// accumulator is rounded nine times per sixteen reduction elements where the fp32 MFMA chain rounds it eight times --
// the same arithmetic class as an fmaf chain, in a different summation order.  NTERM = 6 leaves out a1*b2, a2*b1 and
// a2*b2 (each below 2^-24 of |a*b|: less than the rounding of the accumulator they would be added to).  Nine bf16
// MFMAs cost 9 x 32 cycles per 16 reduction elements against 8 x 64 for fp32 MFMAs: 1.78 x the fp32 matrix peak
// (2.67 x with six terms) at the same LDS read volume per product, with the HBM / L2 traffic of an fp32 kernel.
// (The reference's own fp32 path on its native hardware runs cuDNN convolutions with TF32 inputs -- 10 significand
// bits -- by default, torch.backends.cudnn.allow_tf32; this keeps all 24.)
//
// Pipeline (one workgroup = NW MFMA waves + 4 producer waves, one output tile of BM channels x BN positions):
//   * reduction step ("stage") = one 16-channel group x one tap.  Weights are split and packed ahead of time in HBM in
//     LDS image order wp[phase][m-tile][group][tap][term][h][m][8 ch] (bf16): a stage's slab is contiguous and is
//     copied by ONE producer wave with global_load_lds, 1 KiB per instruction, into a two-slot ring.
//   * the input span of a channel group (all taps read the same span at shifted offsets) is staged by the other three
//     producer waves: 16-byte buffer loads of four consecutive positions x 8 channels per lane (range-checked
//     descriptors: zero padding, ragged tails and channel tails need no predicates), input leaky-ReLU, split into three
//     bf16 terms, three 16-byte LDS writes per position; the loads of the task a wave converts in stage s + 1 are in
//     flight during stage s (the producers pass the stage barrier with `s_waitcnt lgkmcnt(0)` only).  Two span buffers:
//     group g + 1 is staged piecewise during the K stages of group g.
//   * MFMA waves: per stage TM x 3 + TN x 3 `ds_read_b128` fragment reads feed TM x TN x NTERM MFMAs, interleaved over
//     the wave's accumulator tiles so that consecutive MFMAs are independent.
#include <type_traits>

#include "common.h"
#include "prof.h"
#include "conv_tile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// diagnostic build only (-DVCV_X3_STAMPS, tools/probes/x3_stamps.py): 100 MHz stamps of workgroup phases; the product build
// executes none of this
#ifdef VCV_X3_STAMPS
__device__ unsigned long long* g_x3_stamps;
#define VCV_X3_STAMP(k)                                                                                          \
  do {                                                                                                          \
    if (g_x3_stamps && lane == 0)                                                                               \
      g_x3_stamps[((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z))) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define VCV_X3_STAMP(k)
#endif

constexpr int NPROD = 4;  // producer waves: 1 weight-DMA wave + 3 input-staging waves
constexpr int NXW = 3;
constexpr int MAXT = 2;   // pipelined staging tasks per input wave and stage
constexpr int NRING_DEF = 4;  // weight-slab ring slots (a power of two; the 256-row tile takes 2)

// one term a_PA * b_PB of the split product on every accumulator tile of the wave (literal plane indices: a computed
// index makes the compiler select fragment registers at run time)
template <int PA, int PB, int TM, int TN>
__device__ __forceinline__ void mma_term(f32x16 (&acc)[TM][TN], const bf16x8 (&a)[TM][3], const bf16x8 (&b)[TN][3]) {
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
      acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][PA], b[tn][PB], acc[tm][tn], 0, 0, 0);
}

__device__ __forceinline__ void split3(float f, __bf16& t0, __bf16& t1, __bf16& t2) {
  t0 = (__bf16)f;
  const float r1 = f - (float)t0;   // exact
  t1 = (__bf16)r1;
  const float r2 = r1 - (float)t1;  // exact, and representable in 8 bits
  t2 = (__bf16)r2;
}

// ---- weight pack: fp32 w -> split bf16 slabs wp[phase][m-tile][group][j][term][h][m][8] -----------------------------
// mode 0: w is [M, C, K] (forward);  mode 1: w is [C, M, K], A(m, c, j) = w[c, m, K-1-j] (stride-1 data gradient);
// mode 2: w is [C, M, K], residue r = phase keeps taps k = r + j*phases (ConvTranspose forward / strided dgrad)
__global__ void __launch_bounds__(256)
pack_x3_kernel(const float* __restrict__ w, bf16x8* __restrict__ wp, int M, int C, int K, int BM, int JA, int ngr, int nmt,
               int phases, int mode, size_t total) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  size_t t = i;
  const int ml = (int)(t % BM); t /= BM;
  const int hh = (int)(t & 1); t >>= 1;
  const int j = (int)(t % JA); t /= JA;
  const int g = (int)(t % ngr); t /= ngr;
  const int mt = (int)(t % nmt); t /= nmt;
  const int r = (int)t;
  const int m = mt * BM + ml;
  const int c0 = g * 16 + hh * 8;
  bf16x8 v0, v1, v2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c0 + e;
    float f = 0.f;
    if (m < M && c < C) {
      if (mode == 0) { if (j < K) f = w[((size_t)m * C + c) * K + j]; }
      else if (mode == 1) { if (j < K) f = w[((size_t)c * M + m) * K + (K - 1 - j)]; }
      else { const int k = r + j * phases; if (k < K) f = w[((size_t)c * M + m) * K + k]; }
    }
    __bf16 a, b, d;
    split3(f, a, b, d);
    v0[e] = a, v1[e] = b, v2[e] = d;
  }
  // slab (r, mt, g, j): [term][h][m][8]
  const size_t slab = (((size_t)r * nmt + mt) * ngr + g) * JA + j;
  bf16x8* o = wp + slab * (size_t)(3 * 2 * BM) + (size_t)hh * BM + ml;
  o[0] = v0;
  o[(size_t)2 * BM] = v1;
  o[(size_t)4 * BM] = v2;
}

__device__ __forceinline__ void barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BfGeom fields as used here: nch = channel groups of 16, a_bytes = one weight slab (3 * 2 * BM * 16), buf_bytes = one
// span buffer (3 * 2 * xw * 16), JA / phases / ks / vec / ntu / nmt / xw as in conv_pk.hip; BKC = 16, ncg = 1.
// JS: taps per stage (1 or 2).  Two taps double the MFMA work between barriers (the 2 x 1-tile waves of the 128 x 128 tile
// have only 12 MFMAs per tap) at twice the weight-ring footprint; a group's last stage may hold one tap.
template <int NTERM, int TM, int TN, int WM, int WN, bool LEAKY, int NRING = NRING_DEF, int JS = 1>
__global__ void __launch_bounds__(64 * (WM * WN + NPROD))
conv_x3_kernel(const VcvConvArgs p, const BfGeom tg, const char* __restrict__ wp, float* __restrict__ part) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN;
  constexpr int TAPB = 3 * 2 * BM * 16;  // bytes of one tap of a weight slab
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;

  int bxk, mt, r;  // (column-tile, split) index, m-tile, output residue of a phased launch (0 otherwise)
  xcd_tile_id(bxk, mt, r, tg.xcd);
  const int kz = bxk % tg.ks;
  const int bx = bxk / tg.ks;
  const int b = bx / tg.ntu, ut = bx % tg.ntu;
  const int JA = tg.JA, P = p.P, U = p.Q * P, Cg = p.Cg;
  const int K = tg.phases > 1 ? (r < p.K ? (p.K - r + tg.phases - 1) / tg.phases : 0) : p.K;
  const int oo = p.oo + (tg.phases > 1 ? r : 0);
  const int u0 = ut * BN, m0 = mt * BM;
  const int qa = u0 / P;
  const int jspan = (JA - 1) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0;
  const int f0r = (qa * p.s + p.off + jmin) * P;  // first input position the tile reads (flattened [row][P]); may be < 0
  const int f0 = f0r & ~3;                        // the image starts at a multiple of four floats of the channel row
  const int fsh = f0r - f0;
  const int XW = tg.xw;

  const int g_begin = (int)((long long)kz * tg.nch / tg.ks), g_end = (int)((long long)(kz + 1) * tg.nch / tg.ks);
  const int NSG = (K + JS - 1) / JS;      // stages per channel group
  const int S = (g_end - g_begin) * NSG;  // stages
  char* const Aring = smem;
  char* const Xbuf = smem + NRING * tg.a_bytes;

  if (wave >= NW) {
    // ================================================================================================ producers
    __builtin_amdgcn_s_setprio(3);
    if (S <= 0) return;
    const int pw = wave - NW;
    if (pw == 0) {
      VCV_X3_STAMP(4);
      // ---- weight DMA wave: the slab of stage s + NRING - 1 is issued while stage s is multiplied (NRING-slot ring); it
      // passes the barrier of stage s as soon as the slab of stage s + 1 has landed, with later slabs still in flight
      const char* wtile = wp + ((size_t)r * gridDim.y + mt) * (size_t)tg.nch * JA * TAPB;
      constexpr int nT = TAPB >> 10;  // 1 KiB wave-instructions per tap
      constexpr int nA = JS * nT;     // ... per slab: ALWAYS issued in full, so that the in-flight accounting below holds
      auto issue = [&](int g, int sj, int slot) {
        const int j0 = sj * JS;
        char* dst = Aring + slot * tg.a_bytes;
#pragma unroll
        for (int jj = 0; jj < JS; ++jj) {
          const int j = j0 + jj < K ? j0 + jj : j0;  // (a group's odd last tap: its partner slot re-loads the same tap, unread)
          const char* slab = wtile + ((size_t)g * JA + j) * TAPB;
#pragma unroll
          for (int i = 0; i < nT; ++i)
            __builtin_amdgcn_global_load_lds((const void*)(slab + i * 1024 + lane * 16), (lds_ptr)(dst + jj * TAPB + i * 1024), 16, 0, 0);
        }
      };
      int gn = g_begin, jn = 0, issued = 0;
      for (; issued < NRING - 1 && issued < S; ++issued) {
        issue(gn, jn, issued);
        if (++jn == NSG) jn = 0, ++gn;
      }
      // (prologue barrier: slab 0 must be there)
      if (issued >= 3) wait_vm<2 * nA>(); else if (issued == 2) wait_vm<nA>(); else wait_vm<0>();
      asm volatile("s_barrier" ::: "memory");
      for (int s = 0; s < S; ++s) {
        if (issued < S) {
          issue(gn, jn, issued & (NRING - 1));
          if (++jn == NSG) jn = 0, ++gn;
          ++issued;
        }
        // slabs <= s + 1 complete; `issued - (s + 2)` later ones may stay in flight
        const int fly = issued - (s + 2);
        if (fly >= 2) wait_vm<2 * nA>(); else if (fly == 1) wait_vm<nA>(); else wait_vm<0>();
        asm volatile("s_barrier" ::: "memory");
      }
      VCV_X3_STAMP(5);
      return;
    }
    // ---- input staging waves.  A task = (half h of the 16-channel group, block of 256 positions): 8 channels x 4
    // consecutive positions per lane.  Wave xi owns tasks xi, xi + 3, ...; local task i of group g + 1 is converted in
    // stage (g, i mod NSG), and its loads are issued at the START of the stage before (two register sets), so they have a
    // whole stage to arrive.
    const int xi = pw - 1;
    const int npb = (XW + 255) >> 8;
    const int ntask = 2 * npb;
    const int nt_w = ntask > xi ? (ntask - xi + NXW - 1) / NXW : 0;  // tasks of this wave per group
    const long long TinP = (long long)p.Tin * P;
    const float* xb = p.x + (size_t)b * Cg * (size_t)TinP;
    f32x4 xr[2][MAXT][8];
    auto loadT = [&](int g, int i, f32x4 (&dst)[8]) __attribute__((always_inline)) {
      const int task = xi + NXW * i;
      const int hh = task & 1, pb = task >> 1;
      unsigned voff = (unsigned)(f0 + pb * 256 + 4 * lane) * 4u;  // negative -> wraps -> out of range -> 0
      asm volatile("" : "+v"(voff));  // one register (see conv_pk.hip: immediate-offset folding of negative offsets)
      const int c0 = g * 16 + hh * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        const unsigned rec = c < Cg ? (unsigned)(TinP * 4) : 0u;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(xb + (size_t)c * (size_t)TinP), 0, (int)rec, 0x00020000);
        dst[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
      }
    };
    auto storeT = [&](int i, const f32x4 (&src)[8], int buf) __attribute__((always_inline)) {
      const int task = xi + NXW * i;
      const int hh = task & 1, pb = task >> 1;
      const int pos = pb * 256 + 4 * lane;
      if (pos >= XW) return;  // (XW is a multiple of 64: the four positions of a lane are inside or outside together)
      char* base = Xbuf + buf * tg.buf_bytes + ((size_t)hh * XW + pos) * 16;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        bf16x8 v0, v1, v2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = src[e][jj];
          if (LEAKY) f = fmaxf(f, f * p.slope);  // slope in [0, 1)
          __bf16 a, bq, d;
          split3(f, a, bq, d);
          v0[e] = a, v1[e] = bq, v2[e] = d;
        }
        *reinterpret_cast<bf16x8*>(base + (size_t)jj * 16) = v0;
        *reinterpret_cast<bf16x8*>(base + ((size_t)2 * XW + jj) * 16) = v1;
        *reinterpret_cast<bf16x8*>(base + ((size_t)4 * XW + jj) * 16) = v2;
      }
    };
    // loads of the tasks converted in stage (g, j) -- they belong to group g + 1 -- into register set `par`
    auto load_stage = [&](auto par, int g, int j) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par)::value;
      if (g + 1 >= g_end) return;
#pragma unroll
      for (int q = 0; q < MAXT; ++q)
        if (j + q * NSG < nt_w) loadT(g + 1, j + q * NSG, xr[PAR][q]);
    };
    auto store_stage = [&](auto par, int g, int j) __attribute__((always_inline)) {
      constexpr int PAR = decltype(par)::value;
      if (g + 1 >= g_end) return;
      const int nb = (g + 1 - g_begin) & 1;
#pragma unroll
      for (int q = 0; q < MAXT; ++q)
        if (j + q * NSG < nt_w) storeT(j + q * NSG, xr[PAR][q], nb);
      // (more tasks per stage than the pipelined slots hold: load and convert on the spot)
      for (int i = j + MAXT * NSG; i < nt_w; i += NSG) {
        loadT(g + 1, i, xr[PAR][0]);
        storeT(i, xr[PAR][0], nb);
      }
    };
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    // prologue: the whole span of the first group, then the loads of stage 0's tasks
    for (int i = 0; i < nt_w; ++i) {
      loadT(g_begin, i, xr[0][0]);
      storeT(i, xr[0][0], 0);
    }
    int g = g_begin, j = 0;
    load_stage(P0(), g, j);
    if (xi == 0) VCV_X3_STAMP(6);
    barrier_lds();
    for (int s = 0; s < S; s += 2) {
      {  // even stage: its tasks sit in set 0; the next stage's loads go to set 1
        int g2 = g, j2 = j + 1;
        if (j2 == NSG) j2 = 0, ++g2;
        if (s + 1 < S) load_stage(P1(), g2, j2);
        store_stage(P0(), g, j);
        g = g2, j = j2;
        barrier_lds();
      }
      if (s + 1 < S) {  // odd stage
        int g2 = g, j2 = j + 1;
        if (j2 == NSG) j2 = 0, ++g2;
        if (s + 2 < S) load_stage(P0(), g2, j2);
        store_stage(P1(), g, j);
        g = g2, j = j2;
        barrier_lds();
      }
    }
    return;
  }

  // ==================================================================================================== MFMA waves
  if (wave == 0) VCV_X3_STAMP(0);
  const int wm = wave / WN, wn = wave % WN;
  int laneoff[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int u = u0 + (wn * TN + tn) * 32 + l31;
    if (u > U - 1) u = U - 1;
    const int q = u / P, pc = u - q * P;
    laneoff[tn] = (((q - qa) * p.s - jmin) * P + pc + fsh + h * XW) * 16;  // byte offset in a span buffer (h plane included)
  }
  f32x16 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;

  if (S > 0) {
    __syncthreads();
    if (wave == 0) VCV_X3_STAMP(1);
    const int aoff = (h * BM + wm * TM * 32 + l31) * 16;
    int g = g_begin, sj = 0;
    for (int s = 0; s < S; ++s) {
      const char* As0 = Aring + (s & (NRING - 1)) * tg.a_bytes + aoff;
      const int j0 = sj * JS;
#pragma unroll
      for (int jj = 0; jj < JS; ++jj) {
        if (j0 + jj < K) {
          const char* As = As0 + jj * TAPB;
          const char* Xs = Xbuf + ((g - g_begin) & 1) * tg.buf_bytes + (j0 + jj) * p.dj * P * 16;
          bf16x8 a[TM][3], bb[TN][3];
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              a[tm][pl] = *reinterpret_cast<const bf16x8*>(As + ((size_t)pl * 2 * BM + tm * 32) * 16);
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
              bb[tn][pl] = *reinterpret_cast<const bf16x8*>(Xs + laneoff[tn] + (size_t)pl * 2 * XW * 16);
          // small terms first
          if (NTERM == 9) {
            mma_term<2, 2>(acc, a, bb);
            mma_term<1, 2>(acc, a, bb);
            mma_term<2, 1>(acc, a, bb);
          }
          mma_term<0, 2>(acc, a, bb);
          mma_term<2, 0>(acc, a, bb);
          mma_term<1, 1>(acc, a, bb);
          mma_term<0, 1>(acc, a, bb);
          mma_term<1, 0>(acc, a, bb);
          mma_term<0, 0>(acc, a, bb);
        }
      }
      if (++sj == NSG) sj = 0, ++g;
      __syncthreads();  // publishes stage s + 1 (LDS writes + the weight DMA) and retires the reads of stage s
    }
  }
  if (wave == 0) VCV_X3_STAMP(2);
  conv_tile_epilogue<TM, TN>(p, tg, acc, smem, part, wave, wm, wn, lane, b, kz, u0, m0, oo, BM);
  if (wave == 0) VCV_X3_STAMP(3);
}

struct Plan {
  int variant, js;
  int BM, BN, NW;
  BfGeom g;
  size_t scratch_floats, pack_bytes, lds_bytes;
};

bool eligible(const VcvConvArgs& a) {
  const bool fwd_type = a.a_mode == 0 && a.phases <= 1;
  const bool phased = a.a_mode == 1 && a.phases > 1 && a.s == 1 && a.dj == -1;
  return (fwd_type || phased) && a.G == 1 && a.io == 0 && a.post_scale == 0.f && a.ms <= 1 &&
         (a.in_tf == VCV_TF_NONE || (a.in_tf == VCV_TF_LEAKY && a.slope < 1.f && a.slope >= 0.f)) && a.Mg >= 32 &&
         a.Cg >= 16 && a.K <= 16 && a.s >= 1 && a.s <= 3 && (long long)a.Tin * a.P * 4 < (1ll << 31) &&
         (long long)a.Mg * a.Tout * a.P < (1ll << 31);
}

bool make_plan(const VcvConvArgs& a, int BM, int BN, int NW, Plan& pl, int nring = NRING_DEF, int js = 1) {
  pl.BM = BM; pl.BN = BN; pl.NW = NW;
  BfGeom& g = pl.g;
  const int qspan = (BN - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  g.phases = a.phases > 1 ? a.phases : 1;
  g.JA = vcv_cdiv(a.K, g.phases);
  const int rowmax = (qspan * a.s + (g.JA - 1) * adj + 1) * a.P;
  g.xw = (rowmax + 3 + 63) & ~63;  // (up to three floats of round-down at the start)
  g.BKC = 16;
  g.ncg = 1;
  g.nch = vcv_cdiv(a.Cg, 16);
  g.ntu = vcv_cdiv(a.Q * a.P, BN);
  g.nmt = vcv_cdiv(a.Mg, BM);
  g.a_bytes = js * 3 * 2 * BM * 16;  // the weight slab of a stage (js taps)
  pl.js = js;
  g.buf_bytes = 3 * 2 * g.xw * 16;
  pl.lds_bytes = (size_t)nring * g.a_bytes + 2ull * g.buf_bytes;
  if (pl.lds_bytes > VCV_LDS_LIMIT) return false;
  g.ks = 1;
  g.vec = 0;
  const int xcd_remap = vcv_tuning().xcd_remap;
  // nothing to share when a column tile has one workgroup (measured: the re-deal alone costs the 64 x 10 s decode 11 %:
  // eight XCDs walking eight far-apart regions of the tensor instead of one)
  g.xcd = xcd_remap && g.nmt * (g.phases > 1 ? g.phases : 1) > 1;
  pl.pack_bytes = (size_t)g.phases * g.nmt * g.nch * g.JA * (3 * 2 * BM * 16);
  pl.scratch_floats = 0;
  return true;
}

// variants: 0: 128x256 (8 MFMA waves of 2x2 tiles)   1: 128x128 (8 waves of 2x1)   2: 256x128 (8 waves of 2x2)   3: 64x256 (8 waves of 1x2)
//           4: 64x128 (8 waves of 1x1)   5: 32x256 (8 waves of 1x1)   6: 64x512 (8 waves of 1x4)
#define g_force_variant (vcv_tuning().x3_variant)  // tuning probe: -1 = choose
int g_force_js = -1, g_force_ks = -1;

bool choose(const VcvConvArgs& a, Plan& pl) {
  const int U = a.Q * a.P;
  if (U < 96) return false;
  if (g_force_variant >= 0) {  // tools/x3_variant_sweep.py: one fixed tile shape for every launch that can take it
    static const int VBM[7] = {128, 128, 256, 64, 64, 32, 64}, VBN[7] = {256, 128, 128, 256, 128, 256, 512};
    const int v = g_force_variant;
    if (v > 6 || a.Mg < (v == 5 ? 32 : VBM[v] / 2 + 1)) return false;
    const int js = (g_force_js == 2 && (v == 0 || v == 1) && vcv_cdiv(a.K, a.phases > 1 ? a.phases : 1) >= 2) ? 2 : 1;
    if (!make_plan(a, VBM[v], VBN[v], 8, pl, v == 2 ? 2 : NRING_DEF, js)) return false;
    pl.variant = v;
    const int nph0 = a.phases > 1 ? a.phases : 1;
    if (g_force_ks >= 2 && nph0 == 1 && pl.g.nch >= 2 * g_force_ks) {
      pl.g.ks = g_force_ks;
      pl.scratch_floats = (size_t)g_force_ks * a.B * a.Mg * U;
    }
    const bool no_vec0 = !vcv_tuning().pk_vec;
    pl.g.vec = (!no_vec0 && nph0 == 1 && a.os == 1 && a.oo == 0 && (!a.mask || a.P == 1) &&
                (size_t)pl.NW * 32 * 40 * 4 <= pl.lds_bytes) ? 1 : 0;
    return true;
  }
  const int nph = a.phases > 1 ? a.phases : 1;
  auto blocks = [&](int bm, int bn) { return (long long)a.B * vcv_cdiv(U, bn) * vcv_cdiv(a.Mg, bm) * nph; };
  auto eff = [&](int bm, int bn) {
    const long long nb = blocks(bm, bn);
    const long long rounds = (nb + 255) / 256;
    return ((double)U / ((double)vcv_cdiv(U, bn) * bn)) * ((double)a.Mg / ((double)vcv_cdiv(a.Mg, bm) * bm)) *
           ((double)nb / (double)(rounds * 256));
  };
  bool ok = false;
  if (a.Mg >= 96) {
    const double e256 = U > 160 ? eff(128, 256) : 0.0, e128 = eff(128, 128);
    if (e256 >= e128 - 0.02 && e256 > 0.0 && make_plan(a, 128, 256, 8, pl)) pl.variant = 0, ok = true;
    // 128 columns (the strided layers, whose input span is stride x as long, do not fit 256): 256 rows x 128 columns keeps
    // the 2 x 2 accumulator tiles per wave (12 fragment reads per 24 MFMAs instead of 9 per 12) with a two-slot ring
    else if (a.Mg >= 256 && eff(256, 128) >= e128 - 0.1 && make_plan(a, 256, 128, 8, pl, 2)) pl.variant = 2, ok = true;
    else if (make_plan(a, 128, 128, 8, pl)) pl.variant = 1, ok = true;
  } else if (a.Mg >= 48) {
    // 64 x 512 (8 waves of 1 x 4 tiles: 15 fragment reads per 24 MFMAs instead of 9 per 12) where 64 x 256 needs more than one
    // round of 256 workgroups and the wider tile still fills the chip: the generator's 64-channel layers ran two full
    // rounds, each with its own 4 us prologue and 6 us store burst (tools/probes/x3_stamps.py)
    const bool no_v6 = !vcv_tuning().x3_v6;
    if (!no_v6 && nph == 1 && blocks(64, 256) > 256 && blocks(64, 512) >= 192 && eff(64, 512) >= eff(64, 256) - 0.02 &&
        make_plan(a, 64, 512, 8, pl))
      pl.variant = 6, ok = true;
    else if (U > 160 && eff(64, 256) >= eff(64, 128) - 0.02 && make_plan(a, 64, 256, 8, pl)) pl.variant = 3, ok = true;
    else if (make_plan(a, 64, 128, 8, pl)) pl.variant = 4, ok = true;
  } else {
    if (make_plan(a, 32, 256, 8, pl)) pl.variant = 5, ok = true;
  }
  if (!ok) return false;
  // two taps per stage where the doubled weight ring fits (128-row tiles, K >= 2): half the barriers per MFMA
  const bool no_js2 = !vcv_tuning().x3_js2;
  if (!no_js2 && (pl.variant == 0 || pl.variant == 1) && vcv_cdiv(a.K, nph) >= 2) {
    Plan p2;
    if (make_plan(a, pl.BM, pl.BN, pl.NW, p2, NRING_DEF, 2)) { p2.variant = pl.variant; pl = p2; }
  }
  // too few tiles for 256 CUs: split the channel groups over ks workgroups per tile (deterministic slabs + finishing pass)
  const long long nb = blocks(pl.BM, pl.BN);
  if (nph == 1 && nb < 192 && pl.g.nch >= 4) {
    // the split that costs the fewest (rounds of 256 workgroups) x (channel groups per workgroup): rounding to the nearest
    // count put 288 and 260 workgroups -- a second round for 32 and for 4 of them -- on two of DiscriminatorS's last layers
    long long ks = 1, best = (long long)1 << 60;
    const bool old_ks = vcv_tuning().x3_old_ks != 0;
    for (long long c = 2; c <= pl.g.nch / 2; ++c) {
      // (+ 3: a workgroup's prologue and epilogue cost about three channel groups of a 128-row tile; + c: the finishing pass reads c slabs)
      const long long cost = ((nb * c + 255) / 256) * ((pl.g.nch + c - 1) / c + 3) * 64 + c;
      if (cost < best) best = cost, ks = c;
    }
    if (old_ks) ks = (256 + nb / 2) / nb;
    if (ks > pl.g.nch / 2) ks = pl.g.nch / 2;
    if (ks >= 2) {
      pl.g.ks = (int)ks;
      pl.scratch_floats = (size_t)ks * a.B * a.Mg * U;
    }
  }
  const bool no_vec = !vcv_tuning().pk_vec;
  pl.g.vec = (!no_vec && nph == 1 && a.os == 1 && a.oo == 0 && (!a.mask || a.P == 1) &&
              (size_t)pl.NW * 32 * 40 * 4 <= pl.lds_bytes) ? 1 : 0;
  return true;
}

// default: six product terms.  Measured against float64 on the layer shapes of both configs (tests/test_conv_x3_gpu.py,
// profiles/r3_x3_vs_f64.txt) the six- and nine-term results have the same error to three digits, 2e-7 .. 1e-6 of the
// output scale -- the error of the fp32 accumulation order, as large for the fp32-input MFMA kernel -- because each left
// out term is below 2^-24 of its product while one accumulator rounding is 2^-24 of the whole running sum.
#define g_all (vcv_tuning().x3_all)
#define g_terms (vcv_tuning().x3_terms)

template <int NTERM, int TM, int TN, int WM, int WN, int NRING = NRING_DEF, int JS = 1>
int launch(const VcvConvArgs& a, const Plan& pl, char* wp, float* part, int flip, bool pack_valid, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, NT = 64 * (WM * WN + NPROD);
  const BfGeom& g = pl.g;
  if (!pack_valid) {
    const size_t total = (size_t)g.phases * g.nmt * g.nch * g.JA * 2 * BM;
    const int mode = g.phases > 1 ? 2 : (flip ? 1 : 0);
    hipLaunchKernelGGL(pack_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.w, (bf16x8*)wp, a.Mg, a.Cg,
                       a.K, BM, g.JA, g.nch, g.nmt, g.phases, mode, total);
  }
  void (*kern)(const VcvConvArgs, const BfGeom, const char*, float*) =
      a.in_tf == VCV_TF_LEAKY ? conv_x3_kernel<NTERM, TM, TN, WM, WN, true, NRING, JS> : conv_x3_kernel<NTERM, TM, TN, WM, WN, false, NRING, JS>;
  if (pl.lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes) != hipSuccess)
    return VCV_EHIP;
  dim3 grid(a.B * g.ntu * g.ks, g.nmt, g.phases), block(NT);
  const double flops = 2.0 * a.B * a.Mg * a.Cg * a.K * a.P * (double)(g.phases > 1 ? a.Tin : a.Q);
  const int tag[12] = {a.B, 3, a.Cg, a.Mg, a.K, a.Q, a.P, a.s, g.phases, a.a_mode + 10 * g.ks, BM * 1000 + pl.BN, NTERM};
  const double abytes = 4.0 * ((double)a.B * a.Cg * a.Tin * a.P + (double)a.Mg * a.Cg * a.K +
                               (double)a.B * a.Mg * a.Tout * a.P * (1 + (a.res ? 1 : 0) + (a.oaux ? 1 : 0)));
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_CONV_DMA, flops, tag, 12, &ev0, &ev1, abytes, NTERM * flops / VCV_PEAK_BF16_MFMA);
  VCV_LAUNCH_EV(kern, grid, block, (unsigned)pl.lds_bytes, st, ev0, ev1, a, g, (const char*)wp, part);
  if (g.ks > 1) {
    const size_t n = (size_t)a.B * a.Mg * a.Q * a.P;
    if (g.vec && !a.mask && n % 4 == 0 && a.Q == a.Tout && a.Q * a.P >= 4)
      hipLaunchKernelGGL(conv_pk_finish4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, a, (const float*)part, g.ks);
    else
      hipLaunchKernelGGL(conv_pk_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, (const float*)part, g.ks);
  }
  return vcv_check_launch();
}

template <int NTERM>
int run_n(const VcvConvArgs& a, const Plan& pl, char* wp, float* part, int flip, bool pv, hipStream_t st) {
  switch (pl.variant) {
    case 0: return pl.js == 2 ? launch<NTERM, 2, 2, 2, 4, NRING_DEF, 2>(a, pl, wp, part, flip, pv, st)
                              : launch<NTERM, 2, 2, 2, 4>(a, pl, wp, part, flip, pv, st);
    case 1: return pl.js == 2 ? launch<NTERM, 2, 1, 2, 4, NRING_DEF, 2>(a, pl, wp, part, flip, pv, st)
                              : launch<NTERM, 2, 1, 2, 4>(a, pl, wp, part, flip, pv, st);
    case 2: return launch<NTERM, 2, 2, 4, 2, 2>(a, pl, wp, part, flip, pv, st);  // 256 x 128
    case 3: return launch<NTERM, 1, 2, 2, 4>(a, pl, wp, part, flip, pv, st);
    case 4: return launch<NTERM, 1, 1, 2, 4>(a, pl, wp, part, flip, pv, st);
    case 6: return launch<NTERM, 1, 4, 2, 4>(a, pl, wp, part, flip, pv, st);  // 64 x 512
    default: return launch<NTERM, 1, 1, 1, 8>(a, pl, wp, part, flip, pv, st);
  }
}

}  // namespace

// Number of bf16 product terms per fp32 product: 9 (all of them: exact operands) or 6 (the three smallest left out).
extern "C" int vcv_conv_x3_set_terms(int n) {
  if (n != 6 && n != 9) return VCV_EINVAL;
  g_terms = n;
  return VCV_OK;
}
extern "C" int vcv_conv_x3_get_terms(void) { return g_terms; }
// all != 0: take every eligible launch (tests); 0: only the shapes where this kernel is the faster one (wanted())
extern "C" int vcv_conv_x3_set_all(int all) {
  g_all = all ? 1 : 0;
  return VCV_OK;
}
extern "C" int vcv_conv_x3_get_all(void) { return g_all; }
// Tuning probe (tools/x3_variant_sweep.py): fix the tile variant (0..6: 128x256, 128x128, 256x128, 64x256, 64x128, 32x256,
// 64x512; -1 = the library's choice), the taps per stage (2 or 1) and the channel-group split (>= 2, or -1) of every launch.
extern "C" int vcv_conv_x3_set_variant(int variant, int js, int ks) {
  g_force_variant = variant;
  g_force_js = js;
  g_force_ks = ks;
  return VCV_OK;
}

// Same calling convention as vcv_conv_pk_plan / vcv_conv_pk_run: out[0] = size of the packed-weight buffer in 4-byte
// words, out[1] = floats of per-launch scratch, out[2] = signature of the pack layout.
// Where the split kernel is ahead of the fp32-input MFMA kernel (conv_pk.hip) in the training step
// (tools/prof_compare.py on the bench workload): everything but the 32-channel layers and the 64-channel k <= 3 layers,
// whose few reduction stages per tile leave the prologue / epilogue exposed.
static bool wanted(const VcvConvArgs& a) {
  if (g_all) return true;
  if (a.Cg <= 32) return false;
  if (a.Cg <= 64 && a.K <= 3) return false;
  return true;
}

extern "C" int vcv_conv_x3_plan(const VcvConvArgs* args, int flip, int64_t* out) {
  if (!args || !out || !eligible(*args) || !wanted(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose(*args, pl)) return VCV_EINVAL;
  out[0] = (int64_t)((pl.pack_bytes + 3) / 4);
  out[1] = (int64_t)pl.scratch_floats;
  out[2] = ((int64_t)3 << 60) | ((int64_t)pl.BM << 40) | ((int64_t)pl.g.JA << 20) | ((int64_t)pl.g.phases << 8) | (flip ? 1 : 0);
  return 0;
}

extern "C" int vcv_conv_x3_pack_job(const VcvConvArgs* args, int flip, VcvPackJob* out) {
  if (!args || !out || !eligible(*args) || !wanted(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose(*args, pl)) return VCV_EINVAL;
  const BfGeom& g = pl.g;
  out->kind = 0;
  out->M = args->Mg, out->C = args->Cg, out->K = args->K;
  out->BM = pl.BM, out->BKC = 16, out->JA = g.JA, out->nch = g.nch, out->nmt = g.nmt, out->phases = g.phases;
  out->mode = g.phases > 1 ? 2 : (flip ? 1 : 0);
  out->total = (int64_t)g.phases * g.nmt * g.nch * g.JA * 2 * pl.BM;
  return VCV_OK;
}

extern "C" int vcv_conv_x3_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid, void* stream) {
  if (!args || !pack_ws || !eligible(*args) || !wanted(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose(*args, pl)) return VCV_EINVAL;
  if (pl.g.ks > 1 && !scratch_ws) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  return g_terms == 6 ? run_n<6>(*args, pl, (char*)pack_ws, scratch_ws, flip, pack_valid != 0, st)
                      : run_n<9>(*args, pl, (char*)pack_ws, scratch_ws, flip, pack_valid != 0, st);
}

#ifdef VCV_X3_STAMPS
extern "C" int vcv_x3_set_stamps(void* p) {
  unsigned long long* q = (unsigned long long*)p;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_x3_stamps), &q, sizeof(q)) == hipSuccess ? 0 : 1;
}
#endif

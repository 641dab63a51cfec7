// resblock_pair.hip -- one conv PAIR of a HiFi-GAN ResBlock1 as ONE launch over 16-bit activations (inference):
//
//     xt  = leaky( conv1( leaky(x); W1, dilation d ) + b1 )          (k taps, "same" padding)
//     out = conv2( xt; W2, dilation 1 ) + b2 + x                      (k taps, "same" padding)
//     y   = out                      (a middle pair: the next residual stream)
//     y   = y + post_scale * out     (a block's last pair: the stage mean accumulated in place)
//
// Reference: vits/model/modules.py:186-222 (ResBlock1.forward: xt = c1(leaky(x)); xt = c2(leaky(xt)); x = xt + x) behind the
// decoder call of synthesizer_svc.py:108 under fp16 autocast (train.py:104-106).
//
// Why one launch: in the 32- and 64-channel stages of the 48 kHz decode (64 x 10 s: tensors of 2 GB in 16 bits) the two
// launches of a pair move five tensor passes through HBM -- x in, xt out, xt in, x in again (residual), y out -- at
// 2.2 - 3.6 TB/s each and are bound by exactly that (profiles/r4_48k_infer_bf16_*: conv_pk_kernel<Bf16El, ..., IO> 1.4 - 2.0 ms
// per launch).  Here the intermediate xt never leaves the CU: a workgroup stages the input span of its output tile once
// (+ the halo of BOTH convs), keeps leaky(x) and xt as channel-innermost bf16 images in LDS, and the residual re-read of
// x hits L2 a few microseconds after the staging read: two passes instead of five.
//
// Arithmetic = the two-launch path's, rounding for rounding: x (fp16) -> fp32 -> leaky -> bf16 MFMA operand; fp32
// accumulate; + b1, leaky, ONE rounding to bf16 (the stored xt of the two-launch path); fp32 accumulate; + b2 + x, one
// rounding to fp16.  Only the fp32 summation order inside a conv differs (taps outer, channel groups inner).
//
// Layout.  MFMA v_mfma_f32_32x32x16_bf16 (cdna_hip_programming.md section 3): lane l = (r = l & 31, h = l >> 5) holds
// A[row r][k = 8h + j], B[k = 8h + j][col r]; D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h.  Rows = output channels,
// columns = positions, k = 16 input channels of one tap.  LDS images are PLANES of 16-byte slots: plane q holds channels
// 8q .. 8q + 7 of every staged position, so a B fragment is one ds_read_b128 per lane of 32 consecutive positions
// (plane 2 cg + h), and a tap is a position offset.  Slots are rotated inside aligned blocks of 16 positions by the block
// index (slot()): the staging writes -- lane = 8 consecutive positions, one write per position, i.e. a 128-byte lane stride --
// would otherwise land 16 lanes on two banks; with the rotation they are conflict-free and a fragment read of 32
// consecutive positions (which straddles two or three blocks) sees at most a two-way conflict.
// Weights are packed ahead of time (vcv_resblock_pair_pack) as wp[conv][tap][cg][h][m][8]: an A fragment is one
// ds_read_b128 of 32 consecutive rows.
#include "common.h"
#include "prof.h"
#include "conv_tile.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short us8 __attribute__((ext_vector_type(8)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));

constexpr int NWAVE = 8;

__device__ __forceinline__ int slot(int p) { return (p & ~15) | ((p + (p >> 4)) & 15); }

struct PairArgs {
  const unsigned short* x;   // fp16 [B, C, T]
  const bf16x8* wp;          // packed weights of both convs
  const float* b1;
  const float* b2;
  unsigned short* y;         // fp16 [B, C, T]
  int B, T, dil, accumulate;
  float post_scale, slope;
  int ntile;                 // output tiles per batch element
  int dbg;                   // VCVITS_PAIR_DBG (diagnostics): 1 raw conv2 sums, 2 xt read back, 3 staged leaky(x) read back
};

// geometry of one (C, K) instance
template <int C, int K>
struct Geo {
  static constexpr int TM = C / 32;                      // m-tiles (all owned by every wave: B fragments are shared)
  static constexpr int TNW = C == 32 ? 2 : 1;            // n-tiles per wave
  static constexpr int N1 = 32 * NWAVE * TNW;            // xt positions a workgroup computes (conv1 columns)
  static constexpr int H2 = (K - 1) / 2;
  static constexpr int BN = (N1 - 2 * H2) & ~7;          // output positions per workgroup (16-byte rows)
  static constexpr int NQ = C / 8;                       // planes per image
  static constexpr int CG = C / 16;
  static constexpr int WSLOTS = K * CG * 2 * C;          // 16-byte slots of one conv's packed weights
  static constexpr int XT_SLOTS = N1 + 16;               // (columns N1 .. N1 + K - 2 are read for discarded outputs only: never written)
  static constexpr bool BOTHW = (size_t)2 * WSLOTS * 16 <= 56 * 1024;  // both convs' weights resident at once
  // otherwise (64 channels, K >= 7: 57 - 90 KB per conv) the weights are STREAMED: one slab = one tap of one conv
  // ([cg][h][m]: CG * 2 * C slots = 8 KB at C = 64 -- exactly one 16-byte slot per thread of the workgroup) through a ring of
  // NRING slabs, copied three taps ahead by LDS-DMA (conv_mma)
  static constexpr int SLAB = CG * 2 * C;
  static constexpr int NRING = 4;
  static constexpr int W_LDS_SLOTS = BOTHW ? 2 * WSLOTS : NRING * SLAB;
};

template <int C, int K>
__host__ __device__ constexpr int xs_slots(int dil) {
  // staged input positions: N1 + 2 h1 (+ up to 7 of round-down, rounded up to a block of 16, + one block of slack)
  return ((Geo<C, K>::N1 + (K - 1) * dil + 7 + 15) & ~15) + 16;
}

// One conv of the pair on the MFMA pipe: acc[tm][tn] += sum over taps j and 16-channel groups cg of A(j, cg, tm) x B(j, cg, tn),
// B = the image `img` (planes of `pitch` slots) at column (wave's tile column) + j * dstep + shift.
// C = 32: the K * CG weight fragments of the conv stay in registers for the whole tile loop (88 VGPRs at K = 11): per
// MFMA the wave then reads ONE 1 KB fragment from LDS instead of 1.5 -- at eight waves per CU the LDS read rate, not the
// matrix pipe, was the bound.  C = 64 (two m-tiles: 56 - 88 fragments do not fit) reads them per step.
template <int C, int K, int WHICH, int TM, int TNW>
__device__ __forceinline__ void conv_mma(f32x16 (&acc)[TM][TNW], const bf16x8* W, const bf16x8* __restrict__ wpg,
                                         const bf16x8* __restrict__ img, int pitch, int dstep, int shift, int wave, int l31,
                                         int h, int gbase) {
  constexpr int CG = C / 16;
  if constexpr (C == 32) {
    bf16x8 a[K][CG];
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
      for (int cg = 0; cg < CG; ++cg) a[j][cg] = W[((j * CG + cg) * 2 + h) * C + l31];
#pragma unroll
    for (int j = 0; j < K; ++j) {
#pragma unroll
      for (int cg = 0; cg < CG; ++cg) {
        bf16x8 bb[TNW];
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) bb[tn] = img[(2 * cg + h) * pitch + slot((wave * TNW + tn) * 32 + l31 + j * dstep + shift)];
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) acc[0][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][cg], bb[tn], acc[0][tn], 0, 0, 0);
      }
    }
  } else if constexpr (!Geo<C, K>::BOTHW) {
    // streamed weights: W = the ring of four slabs, wpg = the packed weights in global memory, WHICH * K = this conv's first
    // slab.  The ring runs on ACROSS tiles (the 2 K slabs repeat; gbase = ring position of the pair's slab 0).  Tap `sidx`
    // reads ring slot (gbase + sidx) & 3 and starts the LDS-DMA of slab sidx + 3 into the slot the PREVIOUS tap read (free
    // since that tap's barrier); the tap's barrier waits for all but the two youngest DMAs (vmcnt(2)), i.e. for slab
    // sidx + 1.  A copy has two taps of MFMAs to land, takes no registers, and the next tile's first slabs arrive under the
    // epilogue.  The first form loaded slab sidx + 2 into a register and wrote it to the ring in the same tap: the compiler
    // sank the load down to the write -- `global_load; s_waitcnt vmcnt(0); ds_write` at the end of every tap, the whole L2
    // latency exposed 2 K times per tile under a workgroup barrier (MFMA pipe 23 - 27 % busy, 54 % of the wave-cycles in
    // s_waitcnt: profiles/r6_infer_stall_counters.txt) -- plus two exposed loads and a barrier at the head of every tile.
    constexpr int SLAB = Geo<C, K>::SLAB, NR = Geo<C, K>::NRING;
    static_assert(SLAB == 64 * NWAVE, "one 16-byte slot of a slab per thread");
    static_assert(NR == 4, "ring positions are taken & 3; the DMA runs three slabs ahead");
    const int tid = wave * 64 + l31 + 32 * h;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // (Double-buffering the fragments by hand -- the reads of step (j, cg + 1) issued before the MFMAs of step (j, cg) --
    // measured the same 3.85 ms at K = 11: with A fragments re-read by all eight waves the tap moves 96 KB through an LDS
    // that passes 128 B / clock, 768 clocks against 512 of MFMA work per SIMD; the read LATENCY is not the bound.)
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
      const int sidx = WHICH * K + j;
      const int cur = gbase + sidx;
      const int nid = sidx + 3 < 2 * K ? sidx + 3 : sidx + 3 - 2 * K;
      __builtin_amdgcn_global_load_lds((const void*)(wpg + nid * SLAB + tid),
                                       (lds_ptr)(const_cast<bf16x8*>(W) + ((cur + 3) & (NR - 1)) * SLAB + wave * 64), 16, 0, 0);
      const bf16x8* Wj = W + (cur & (NR - 1)) * SLAB;
#pragma unroll
      for (int cg = 0; cg < CG; ++cg) {
        bf16x8 a[TM], bb[TNW];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = Wj[(cg * 2 + h) * C + tm * 32 + l31];
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) bb[tn] = img[(2 * cg + h) * pitch + slot((wave * TNW + tn) * 32 + l31 + j * dstep + shift)];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TNW; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      }
      // all but the two youngest DMAs have landed: slab sidx + 1 (this wave's part; the barrier makes it everybody's), and
      // every wave is past its reads of slab sidx
      asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
#pragma unroll 1
    for (int j = 0; j < K; ++j) {
#pragma unroll
      for (int cg = 0; cg < CG; ++cg) {
        bf16x8 a[TM], bb[TNW];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = W[((j * CG + cg) * 2 + h) * C + tm * 32 + l31];
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn) bb[tn] = img[(2 * cg + h) * pitch + slot((wave * TNW + tn) * 32 + l31 + j * dstep + shift)];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TNW; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bb[tn], acc[tm][tn], 0, 0, 0);
      }
    }
  }
}

// PERSISTENT workgroups: the grid is one workgroup per CU-slot; each loads the packed weights ONCE (where both convs' fit: at
// K = 11 they are 45 KB -- more than the 36 KB of activations a tile reads, and the first version of this kernel, one
// workgroup per tile, spent its time re-loading them and waiting on serialised phases: 2.5 ms per pair) and then walks tiles
// tile0, tile0 + grid, ...  The global loads of the NEXT tile's input span are issued before the current tile's two convs and
// land in registers while the matrix pipe works; they are converted and written to the LDS image after the current tile's
// epilogue.  Staging task = (plane q, 8 consecutive positions) with q uniform per wave (the row's buffer descriptor is a
// scalar operand: a per-lane q would put a waterfall loop around every load).
template <int C, int K>
__global__ void __launch_bounds__(64 * NWAVE)
resblock_pair_kernel(const PairArgs p, const int xs_n, const int total_tiles) {
  using G = Geo<C, K>;
  constexpr int TM = G::TM, TNW = G::TNW, N1 = G::N1, H2 = G::H2, BN = G::BN, NQ = G::NQ;
  constexpr int WPQ = NWAVE / NQ;  // waves that share one plane's staging (C = 32: 2, C = 64: 1)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16x8* Ws = reinterpret_cast<bf16x8*>(smem);                       // [1 or 2][WSLOTS]
  bf16x8* Xs = Ws + G::W_LDS_SLOTS;                                    // [NQ][xs_n]   (later: the waves' epilogue tiles)
  const int xs_bytes = NQ * xs_n * 16;
  const int ep_bytes = NWAVE * 32 * 40 * 4;
  bf16x8* XTs = reinterpret_cast<bf16x8*>(reinterpret_cast<char*>(Xs) + (xs_bytes > ep_bytes ? xs_bytes : ep_bytes));  // [NQ][XT_SLOTS]
  // the tile's x as it is in HBM (fp16, position-innermost rows of BN): the epilogue's residual operand.  Re-reading it from
  // global memory cost a third HBM pass (rocprofv3 FETCH_SIZE: 5.1 GB read per launch against 1.97 GB of x + halo: the
  // re-read missed L2 more often than not)
  unsigned short* Xraw = reinterpret_cast<unsigned short*>(XTs + NQ * G::XT_SLOTS);  // [C][BN]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int T = p.T, dil = p.dil;
  const int h1 = (K - 1) * dil / 2;
  const int npg = xs_n / 8;
  const int sq = wave % NQ;                       // the plane this wave stages (wave-uniform)
  const int spg = lane + 64 * (wave / NQ);        // ... and the lane's position group (npg <= 64 * WPQ: one task per lane)
  const bool stask = spg < npg;
  const float sl = p.slope;

  u32x4 v[8];
  auto issue = [&](int tile) {  // global loads of tile's input span -> v[]
    const int b = tile / p.ntile, t0 = (tile - b * p.ntile) * BN;
    const int ps8 = (t0 - H2 - h1) & ~7;
    const int pos0 = ps8 + 8 * spg;  // global position of the lane's first element (a multiple of 8)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const unsigned short* row = p.x + ((size_t)b * C + (sq * 8 + e)) * (size_t)T;
      __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, 0, T * 2, 0x00020000);
      // before the row: wraps -> out of range -> zeros (rows are multiples of 8 elements); a lane without a task: out of range too
      unsigned voff = stask ? (unsigned)pos0 * 2u : 0xffffff00u;
      asm volatile("" : "+v"(voff));
      v[e] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    }
  };
  auto stage = [&](int tile) {  // v[] -> leaky, bf16 -> the Xs image; the raw central part -> Xraw
    if (!stask) return;
    {
      const int bb = tile / p.ntile, tt0 = (tile - bb * p.ntile) * BN;
      const int o = 8 * spg - (tt0 - ((tt0 - H2 - h1) & ~7));  // output position of the lane's first element (a multiple of 8)
      if (o >= 0 && o < BN) {
#pragma unroll
        for (int e = 0; e < 8; ++e) *reinterpret_cast<u32x4*>(Xraw + (sq * 8 + e) * BN + o) = v[e];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // position i of the lane = halfword i & 1 of dword i >> 1 of each channel's load (extracted from the dwords: a
        // __builtin_bit_cast(_Float16, vec[i]) on a 16-bit vector ELEMENT expression compiled to element 0 for every i)
        const unsigned short bits = (unsigned short)((i & 1) ? (v[e][i >> 1] >> 16) : (v[e][i >> 1] & 0xffffu));
        const float f = (float)__builtin_bit_cast(_Float16, bits);
        o[e] = (__bf16)fmaxf(f, f * sl);  // leaky (0 <= slope < 1), one rounding to the bf16 operand
      }
      Xs[sq * xs_n + slot(8 * spg + i)] = o;
    }
  };

  int tile = blockIdx.x;
  if (tile >= total_tiles) return;
  issue(tile);
  int gbase = 0;  // streamed weights: ring position of the pair's slab 0 (conv_mma)
  if (G::BOTHW) {
    for (int i = tid; i < 2 * G::WSLOTS; i += 64 * NWAVE) Ws[i] = p.wp[i];
  } else {  // the first three slabs; from here on the ring feeds itself, across tiles
    Ws[tid] = p.wp[tid];
    Ws[G::SLAB + tid] = p.wp[G::SLAB + tid];
    Ws[2 * G::SLAB + tid] = p.wp[2 * G::SLAB + tid];
  }
  stage(tile);
  __syncthreads();

  for (;;) {
    const int b = tile / p.ntile, t0 = (tile - b * p.ntile) * BN;
    const int sh = (t0 - H2 - h1) & 7;
    const int next = tile + (int)gridDim.x;
    const bool more = next < total_tiles;
    if (more) issue(next);  // in flight under this tile's two convs

    // ---- conv1: xt[m][n], n in [0, N1): wave w owns n-tiles w * TNW .. ----
    f32x16 acc[TM][TNW];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;
    conv_mma<C, K, 0>(acc, Ws, p.wp, Xs, xs_n, dil, sh, wave, l31, h, gbase);
    // epilogue 1: + b1, leaky, bf16 -> XTs; columns outside [0, T) are the zero padding of conv2's input
#pragma unroll
    for (int tn = 0; tn < TNW; ++tn) {
      const int n = (wave * TNW + tn) * 32 + l31;
      const int gp = t0 - H2 + n;
      const bool inside = gp >= 0 && gp < T;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {  // registers 4 qq .. 4 qq + 3: channels tm * 32 + 8 qq + 4 h + (0 .. 3)
          us4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int m = tm * 32 + 8 * qq + 4 * h + r;
            float val = acc[tm][tn][4 * qq + r] + p.b1[m];
            val = fmaxf(val, val * sl);
            o[r] = inside ? __builtin_bit_cast(unsigned short, (__bf16)val) : (unsigned short)0;
          }
          // plane = channels / 8 = tm * 4 + qq; the lane's four channels are the low (h = 0) or high (h = 1) half of the slot
          us4* dst = reinterpret_cast<us4*>(XTs + (tm * 4 + qq) * G::XT_SLOTS + slot(n)) + h;
          *dst = o;
        }
      }
    }
    __syncthreads();

    // a block's last pair: the accumulate target is requested NOW, so its latency hides under conv2 (K <= 7; at K = 11 the
    // 22 weight fragments + these registers cost more than the latency: measured 2.21 -> 2.57 ms, so K = 11 loads it late)
    constexpr bool PRE = K <= 7;
    us8 yacc[TNW][TM][2];
    if constexpr (PRE)
      if (p.accumulate)
#pragma unroll
        for (int tn = 0; tn < TNW; ++tn)
#pragma unroll
          for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
              const int m = tm * 32 + ps * 16 + (lane >> 2);
              const int o = (wave * TNW + tn) * 32 + 8 * (lane & 3);
              const bool ok = o < BN && t0 + o < T;
              const size_t idx = ((size_t)b * C + m) * (size_t)T + (ok ? t0 + o : 0);
              yacc[tn][tm][ps] = *reinterpret_cast<const us8*>(p.y + idx);
            }

    // ---- conv2: out[m][o], o in [0, BN): same tile ownership; xt column of (o, tap j) = o + j ----
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;
    conv_mma<C, K, 1>(acc, Ws + (G::BOTHW ? G::WSLOTS : 0), p.wp, XTs, G::XT_SLOTS, 1, 0, wave, l31, h, gbase);
    gbase = (gbase + 2 * K) & 3;
    // epilogue 2 through the wave's LDS tile (the Xs region: nobody reads it after conv1's barrier): rows of eight consecutive
    // positions per lane, + b2 + x (fp16, re-read: L2), [y += post_scale * out], one rounding to fp16, 16-byte stores
    {
      float* Tl = reinterpret_cast<float*>(Xs) + wave * (32 * 40);
      const float ps_ = p.post_scale != 0.f ? p.post_scale : 1.f;
#pragma unroll
      for (int tn = 0; tn < TNW; ++tn) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
          if (p.dbg < 3)
#pragma unroll
            for (int e = 0; e < 16; ++e) Tl[((e & 3) + 8 * (e >> 2) + 4 * h) * 40 + l31] = acc[tm][tn][e];
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int r = ps * 16 + (lane >> 2), c8 = lane & 3;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(Tl + r * 40 + 8 * c8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(Tl + r * 40 + 8 * c8 + 4);
            const int m = tm * 32 + r;
            const int o = (wave * TNW + tn) * 32 + 8 * c8;
            const int t = t0 + o;
            if (o >= BN || t >= T) continue;  // (BN and T are multiples of 8: a group of eight is all in or all out)
            const size_t idx = ((size_t)b * C + m) * (size_t)T + t;
            if (p.dbg) {
              us8 out;
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                float dv = i < 4 ? a0[i] : a1[i - 4];
                if (p.dbg == 2) dv = (float)(reinterpret_cast<const __bf16*>(XTs + (m >> 3) * G::XT_SLOTS + slot(o + i + H2))[m & 7]);
                if (p.dbg == 3) dv = (float)(reinterpret_cast<const __bf16*>(Xs + (m >> 3) * xs_n + slot(o + i + H2 + h1 + sh))[m & 7]);
                out[i] = f32_to_us<2>(dv);
              }
              *reinterpret_cast<us8*>(p.y + idx) = out;
              continue;
            }
            const us8 xr = *reinterpret_cast<const us8*>(Xraw + m * BN + o);  // the residual, from the staged tile
            float val[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            float yy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (p.accumulate) {
              us8 y8;
              if constexpr (PRE) y8 = yacc[tn][tm][ps];
              else y8 = *reinterpret_cast<const us8*>(p.y + idx);
#pragma unroll
              for (int i = 0; i < 8; ++i) yy[i] = us_to_f32(y8[i], 2);
            }
            const float bv = p.b2[m];
            us8 out;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              float xv = val[i] + bv + us_to_f32(xr[i], 2);  // (through the helper: see stage())
              xv = xv * ps_ + yy[i];
              out[i] = f32_to_us<2>(xv);
            }
            *reinterpret_cast<us8*>(p.y + idx) = out;
          }
        }
      }
    }
    if (!more) break;
    __syncthreads();  // every wave's epilogue tile (the Xs region) is done
    stage(next);      // the next tile's image
    __syncthreads();
    tile = next;
  }
}

// fp32 [C][C][K] x 2 -> wp[conv][tap][cg][h][m][8] (bf16)
__global__ void __launch_bounds__(256)
resblock_pair_pack_kernel(const float* __restrict__ w1, const float* __restrict__ w2, bf16x8* __restrict__ wp, int C, int K) {
  const int per = K * (C / 16) * 2 * C;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * per) return;
  const int conv = i / per;
  int t = i - conv * per;
  const int m = t % C; t /= C;
  const int hh = t & 1; t >>= 1;
  const int cg = t % (C / 16);
  const int j = t / (C / 16);
  const float* w = conv ? w2 : w1;
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)w[((size_t)m * C + cg * 16 + hh * 8 + e) * K + j];
  wp[i] = v;
}

template <int C, int K>
size_t lds_bytes(int dil) {
  using G = Geo<C, K>;
  const size_t xs = (size_t)G::NQ * xs_slots<C, K>(dil) * 16, ep = (size_t)NWAVE * 32 * 40 * 4;
  return (size_t)G::W_LDS_SLOTS * 16 + (xs > ep ? xs : ep) + (size_t)G::NQ * G::XT_SLOTS * 16 + (size_t)C * G::BN * 2;
}

template <int C, int K>
int launch(const VcvResPairArgs& a, hipStream_t st) {
  using G = Geo<C, K>;
  const size_t lds = lds_bytes<C, K>(a.dil);
  if (lds > VCV_LDS_LIMIT) return VCV_EINVAL;
  auto kern = resblock_pair_kernel<C, K>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VCV_EHIP;
  PairArgs p;
  p.x = (const unsigned short*)a.x; p.wp = (const bf16x8*)a.wp; p.b1 = a.b1; p.b2 = a.b2; p.y = (unsigned short*)a.y;
  p.B = a.B; p.T = a.T; p.dil = a.dil; p.accumulate = a.accumulate; p.post_scale = a.post_scale; p.slope = a.slope;
  p.ntile = vcv_cdiv(a.T, G::BN);
  p.dbg = vcv_tuning().pair_dbg;
  const long long nblk = (long long)a.B * p.ntile;
  if (nblk >= (1ll << 31)) return VCV_EINVAL;
  // persistent: as many workgroups as the chip holds at this LDS footprint (256 CUs; two per CU where two fit)
  const long long slots = 256 * (2 * lds <= VCV_LDS_LIMIT ? 2 : 1);
  const long long forced = vcv_tuning().pair_grid;
  const long long grid = forced > 0 ? (forced < nblk ? forced : nblk) : (nblk < slots ? nblk : slots);
  if (xs_slots<C, K>(a.dil) / 8 > 64 * (NWAVE / G::NQ)) return VCV_EINVAL;  // (one staging task per lane)
  // per-launch events of bench.py's roofline object: the packed-weight conv class (the launch replaces two of its members)
  const double flops = 2.0 * 2.0 * a.B * (double)C * C * K * (double)a.T;
  const double abytes = 2.0 * (double)a.B * C * (double)a.T * (2 + (a.accumulate ? 1 : 0)) + 2.0 * 2.0 * C * C * K;
  const int tag[12] = {a.B, 2, C, C, K, a.T, 1, 1, 1, 100 + a.dil, C * 1000 + G::BN, 16};
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_CONV_DMA, flops, tag, 12, &ev0, &ev1, abytes, flops / VCV_PEAK_BF16_MFMA);
  VCV_LAUNCH_EV(kern, dim3((unsigned)grid), dim3(64 * NWAVE), (unsigned)lds, st, ev0, ev1, p, xs_slots<C, K>(a.dil), (int)nblk);
  return vcv_check_launch();
}

bool supported(int C, int K, int dil, int T) {
  if ((C != 32 && C != 64) || (K != 3 && K != 7 && K != 11) || dil < 1 || dil > 5 || T < 64 || (T & 7)) return false;
  // 64 channels with K >= 7: the two convs' weights (114 KB at K = 7) do not fit next to the images; re-loading them per tile
  // made the fused launch no faster than the two it replaces (3.5 vs 3.5 - 3.8 ms on the 48 kHz decode), so they are streamed
  // tap by tap through a ring under the MFMAs (conv_mma)
  const bool no_stream = !vcv_tuning().pair_stream;  // (A/B: leave 64 channels x K >= 7 to the two launches)
  if (C == 64 && K > 3 && no_stream) return false;
  if ((long long)T * 2 >= (1ll << 31)) return false;
  size_t lds = 0;
  if (C == 32) lds = K == 3 ? lds_bytes<32, 3>(dil) : K == 7 ? lds_bytes<32, 7>(dil) : lds_bytes<32, 11>(dil);
  else lds = K == 3 ? lds_bytes<64, 3>(dil) : K == 7 ? lds_bytes<64, 7>(dil) : lds_bytes<64, 11>(dil);
  return lds <= VCV_LDS_LIMIT;
}

}  // namespace

// Bytes of the packed weight buffer of a (C, K, dil, T) pair, or 0 when the fused kernel does not take the shape (the caller
// runs the pair as two vcv_conv_bf16io_* launches).
extern "C" int64_t vcv_resblock_pair_supported(int C, int K, int dil, int T) {
  return supported(C, K, dil, T) ? (int64_t)2 * K * C * C * 2 : 0;
}

extern "C" int vcv_resblock_pair_pack(const float* w1, const float* w2, void* wp, int C, int K, void* stream) {
  if (!w1 || !w2 || !wp || (C != 32 && C != 64) || K < 1 || K > 16) return VCV_EINVAL;
  const int n = 2 * K * (C / 16) * 2 * C;
  hipLaunchKernelGGL(resblock_pair_pack_kernel, dim3((unsigned)vcv_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, w1, w2,
                     (bf16x8*)wp, C, K);
  return vcv_check_launch();
}

extern "C" int vcv_resblock_pair_x16(const VcvResPairArgs* a, void* stream) {
  if (!a || !a->x || !a->wp || !a->b1 || !a->b2 || !a->y || a->B <= 0 || !supported(a->C, a->K, a->dil, a->T)) return VCV_EINVAL;
  if ((((uintptr_t)a->x | (uintptr_t)a->y | (uintptr_t)a->wp) & 15) || a->slope < 0.f || a->slope >= 1.f) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (a->C == 32) {
    if (a->K == 3) return launch<32, 3>(*a, st);
    if (a->K == 7) return launch<32, 7>(*a, st);
    return launch<32, 11>(*a, st);
  }
  if (a->K == 3) return launch<64, 3>(*a, st);
  if (a->K == 7) return launch<64, 7>(*a, st);
  return launch<64, 11>(*a, st);
}

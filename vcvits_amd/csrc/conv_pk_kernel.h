// conv_pk_kernel.h -- the packed-operand implicit-GEMM convolution kernel template, its tile planner and launcher,
// shared by conv_pk.hip (fp32 activations in HBM: vcv_conv_bf16_* / vcv_conv_pk_*) and conv_pk_io.hip (bf16 activations
// in HBM: vcv_conv_bf16io_*).  See conv_pk.hip for the design notes.
#pragma once
#include "common.h"
#include "prof.h"
#include "conv_tile.h"  // BfGeom, conv_tile_epilogue, conv_pk_finish*_kernel

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr;

// Element traits.  The kernel moves 16-byte channel groups: 8 bf16 channels (one v_mfma_f32_32x32x16_bf16 per pair of
// fragments) or 4 fp32 channels (four v_mfma_f32_32x32x2_f32: MFMA step i takes channel i of the h = 0 lanes' group
// and channel i of the h = 1 lanes' group -- the reduction order is free as long as both operands agree).
struct Bf16El {
  typedef bf16x8 frag;
  typedef __bf16 elem;
  static constexpr int CPG = 8;   // channels per 16-byte group
  static constexpr int ESZ = 2;
  static __device__ __forceinline__ void set(frag& v, int e, float f) { v[e] = (__bf16)f; }
  static __device__ __forceinline__ f32x16 mma(const frag& a, const frag& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
struct F32El {
  typedef f32x4 frag;
  typedef float elem;
  static constexpr int CPG = 4;
  static constexpr int ESZ = 4;
  static __device__ __forceinline__ void set(frag& v, int e, float f) { v[e] = f; }
  static __device__ __forceinline__ f32x16 mma(const frag& a, const frag& b, f32x16 c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c, 0, 0, 0);
    return c;
  }
};


// ---- weight pack: fp32 w -> bf16 slabs wp[phase][m-tile][chunk][j][cg][h][m][8] -------------------------------
// mode 0: w is [M, C, K] (forward);  mode 1: w is [C, M, K], A(m, c, j) = w[c, m, K-1-j] (stride-1 data gradient);
// mode 2: w is [C, M, K], residue r = phase keeps taps k = r + j*phases (ConvTranspose forward / strided dgrad)
template <class EL>
__global__ void __launch_bounds__(256)
pack_pk_kernel(const float* __restrict__ w, typename EL::frag* __restrict__ wp, int M, int C, int K, int BM, int BKC, int JA,
                 int nch, int nmt, int phases, int mode, size_t total) {
  constexpr int CPG = EL::CPG;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ncg = BKC / (2 * CPG);
  size_t t = i;
  const int ml = (int)(t % BM); t /= BM;
  const int hh = (int)(t & 1); t >>= 1;
  const int cg = (int)(t % ncg); t /= ncg;
  const int j = (int)(t % JA); t /= JA;
  const int ch = (int)(t % nch); t /= nch;
  const int mt = (int)(t % nmt); t /= nmt;
  const int r = (int)t;
  const int m = mt * BM + ml;
  const int c0 = ch * BKC + cg * 2 * CPG + hh * CPG;
  typename EL::frag v;
#pragma unroll
  for (int e = 0; e < CPG; ++e) {
    const int c = c0 + e;
    float f = 0.f;
    if (m < M && c < C) {
      if (mode == 0) { if (j < K) f = w[((size_t)m * C + c) * K + j]; }
      else if (mode == 1) { if (j < K) f = w[((size_t)c * M + m) * K + (K - 1 - j)]; }
      else if (mode == 2) { const int k = r + j * phases; if (k < K) f = w[((size_t)c * M + m) * K + k]; }
      // mode 3 (merged phases): row m = (cout, phase) with the phase fastest; M / K here are the launch's (Cout * phases rows,
      // taps per phase); w is [C, Cout, K * phases]
      else { const int co = m / phases, ph = m - co * phases; if (j < K) f = w[((size_t)c * (M / phases) + co) * (K * phases) + ph + j * phases]; }
    }
    EL::set(v, e, f);
  }
  wp[i] = v;
}

// NP = 0: every wave stages and multiplies.  NP > 0: warp-specialised -- the NP waves after the NW MFMA waves do all
// the staging (weight DMA, input loads and LDS writes of chunk c+1 while the MFMA waves multiply chunk c), so the
// MFMA waves never sit in the vector-memory issue queue (see wgrad_dma.hip for the measurements behind this).
// X4: a staging task is a 16-byte channel group x 256 positions, each lane loading FOUR consecutive positions of every
// channel with one 16-byte buffer load (a 4x4 block that is transposed by register naming: a quarter of the load
// instructions); otherwise x 64 positions with one dword load per channel.
// IO (bit 0: `x` is a 16-bit tensor in HBM; bit 1: `y`, `res` and the accumulate target are; bit 2 / bit 3: that `x` / `y` is
// fp16, else bf16; 0 in conv_pk.hip, the 16-bit combinations in conv_pk_io*.hip): a 16-bit `x` is staged in tasks of a 16-byte channel group x 512 positions, each lane loading EIGHT consecutive
// positions of every channel with one 16-byte buffer load (rows of an even number of elements: 4-byte aligned); the 8 x 8
// block is transposed by register naming and goes out as eight 16-byte LDS writes -- no conversion unless the input
// leaky-ReLU is fused (then through fp32 and back, the rounding the fp32-activation path applies).
template <class EL, int TM, int TN, int WM, int WN, bool LEAKY, int MAXT, int NP = 0, bool X4 = false, int IO = 0>
__global__ void __launch_bounds__(64 * (WM * WN + NP))
conv_pk_kernel(const VcvConvArgs p, const BfGeom tg, const typename EL::frag* __restrict__ wp, float* __restrict__ part) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN;
  constexpr int NS = NP ? NP : NW;  // staging waves
  constexpr int CPG = EL::CPG;
  constexpr bool XB = (IO & 1) != 0;
  constexpr bool XF16 = (IO & 4) != 0;  // the 16-bit `x` is fp16 (else bf16)
  static_assert(!XB || (X4 && EL::ESZ == 2), "bf16 activations: the bf16 element type, 16-byte loads");
  constexpr int PPL = XB ? 8 : (X4 ? 4 : 1);  // consecutive positions a lane loads per channel
  typedef typename EL::frag frag;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, h = lane >> 5;
  const int wm = wave / WN, wn = wave % WN;
  const int sw = NP ? wave - NW : wave;  // index among the staging waves (negative: an MFMA wave of a specialised launch)

  int bxk, mt, r;  // (column-tile, split) index, m-tile, output residue of a phased launch (0 otherwise)
  // (16-bit-activation kernels: plain order, compiled without the re-deal -- measured on the 64 x 10 s decode the mere
  // presence of its code, switched off at run time, cost them 10 %: RTF 0.000205 vs 0.000186, same box, profiles/r4_xcd_ab.txt)
  if constexpr (IO != 0) bxk = blockIdx.x, mt = blockIdx.y, r = blockIdx.z;
  else xcd_tile_id(bxk, mt, r, tg.xcd);
  const int kz = bxk % tg.ks;
  const int bx = bxk / tg.ks;
  const int b = bx / tg.ntu, ut = bx % tg.ntu;
  const int JA = tg.JA, P = p.P, U = p.Q * P, Cg = p.Cg, Mg = p.Mg;
  const int K = tg.phases > 1 ? (r < p.K ? (p.K - r + tg.phases - 1) / tg.phases : 0) : p.K;
  const int oo = p.oo + (tg.phases > 1 ? r : 0);
  const int u0 = ut * BN, m0 = mt * BM;
  const int qa = u0 / P;
  const int jspan = (JA - 1) * p.dj;
  const int jmin = jspan < 0 ? jspan : 0;
  const int f0r = (qa * p.s + p.off + jmin) * P;  // first input position the tile reads (flattened [row][P]); may be < 0
  // X4: the image starts at that rounded down to a multiple of four floats of the channel row, so that no 16-byte load
  // straddles the row start (a load that begins before the buffer comes back as zeros as a whole)
  const int f0 = PPL > 1 ? (f0r & ~(PPL - 1)) : f0r;
  const int fsh = f0r - f0;
  const int BKC = tg.BKC, ncg = tg.ncg, XW = tg.xw;

  int laneoff[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    int u = u0 + (wn * TN + tn) * 32 + l31;
    if (u > U - 1) u = U - 1;
    const int q = u / P, pc = u - q * P;
    laneoff[tn] = (((q - qa) * p.s - jmin) * P + pc + fsh + h * XW) * 16;  // byte offset in the Xs image (h plane included)
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
  const char* wtile = (const char*)wp + ((size_t)r * gridDim.y + mt) * tg.nch * (size_t)tg.a_bytes;
  const int nA = tg.a_bytes >> 10;  // 1 KiB wave-instructions per weight slab
  const int npb = PPL > 1 ? (XW + 64 * PPL - 1) / (64 * PPL) : XW >> 6;  // position blocks per span (64 * PPL positions)
  const int ntask = (BKC / CPG) * npb;  // (16-byte channel group, position block) staging tasks per chunk

  auto issueA = [&](int ch, int buf) {
    char* As = smem + buf * tg.buf_bytes;
    const char* slab = wtile + (size_t)ch * tg.a_bytes;
    for (int i = sw; i < nA; i += NS)
      __builtin_amdgcn_global_load_lds((const void*)(slab + i * 1024 + lane * 16), (lds_ptr)(As + i * 1024), 16, 0, 0);
  };
  // registers of the input loads in flight: task t of this wave = staging task sw + t * NS
  float xr[MAXT][X4 ? 4 * CPG : CPG];  // (XB: 8 channels x 4 dwords of eight bf16 positions)
  auto loadX = [&](int ch, int tbase = 0) {
    const int c0 = ch * BKC;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int task = tbase + sw + t * NS;
      if (task < ntask) {
        const int g8 = task / npb, pb = task - g8 * npb;
        unsigned voff = XB ? (unsigned)(f0 + pb * 512 + 8 * lane) * 2u
                           : (unsigned)(X4 ? f0 + pb * 256 + 4 * lane : f0 + pb * 64 + lane) * 4u;  // negative -> wraps -> out of range -> 0
        // (opaque to the compiler: it otherwise moves the constant part of the task index into the instruction's
        // immediate offset, and a negative register offset plus an immediate that sum to 0 or 4 came back as zero)
        asm volatile("" : "+v"(voff));
#pragma unroll
        for (int e = 0; e < CPG; ++e) {
          const int c = c0 + g8 * CPG + e;
          const unsigned rec = c < Cg ? (unsigned)(TinP * (XB ? 2 : 4)) : 0u;
          __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
              XB ? (void*)((const char*)p.x + ((size_t)b * Cg + c) * (size_t)TinP * 2) : (void*)(xb + (size_t)c * (size_t)TinP), 0, (int)rec, 0x00020000);
          if constexpr (XB) {
            const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) xr[t][(X4 ? 4 : 1) * e + (X4 ? jj : 0)] = v4[jj];
          } else if (X4) {
            const f32x4 v4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) xr[t][(X4 ? 4 : 1) * e + (X4 ? jj : 0)] = v4[jj];
          } else {
            xr[t][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
          }
        }
      }
    }
  };
  auto storeX = [&](int buf, int tbase = 0) {
    char* Xs = smem + buf * tg.buf_bytes + tg.a_bytes;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
      const int task = tbase + sw + t * NS;
      if (task < ntask) {
        const int g8 = task / npb, pb = task - g8 * npb;
        if constexpr (XB) {
          const int pos = pb * 512 + 8 * lane;
          if (pos < XW) {  // (XW is a multiple of 64: the eight positions of a lane are inside or outside together)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
              // position jj of the lane: one dword per channel pair (2 e2, 2 e2 + 1), taken from halfword jj & 1 of dword
              // jj >> 1 of each channel's load
              typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
              typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
              u32x4 d;
#pragma unroll
              for (int e2 = 0; e2 < 4; ++e2) {
                const unsigned lo = __float_as_uint(xr[t][4 * (2 * e2) + (jj >> 1)]);
                const unsigned hi = __float_as_uint(xr[t][4 * (2 * e2 + 1) + (jj >> 1)]);
                if (LEAKY || XF16) {
                  float f0, f1;
                  if constexpr (XF16) {  // fp16 storage: through fp32 to the bf16 MFMA operand
                    f0 = (float)__builtin_bit_cast(_Float16, (unsigned short)((jj & 1) ? (lo >> 16) : (lo & 0xffffu)));
                    f1 = (float)__builtin_bit_cast(_Float16, (unsigned short)((jj & 1) ? (hi >> 16) : (hi & 0xffffu)));
                  } else {
                    f0 = __uint_as_float((jj & 1) ? (lo & 0xffff0000u) : (lo << 16));
                    f1 = __uint_as_float((jj & 1) ? (hi & 0xffff0000u) : (hi << 16));
                  }
                  if (LEAKY) {
                    f0 = fmaxf(f0, f0 * p.slope);
                    f1 = fmaxf(f1, f1 * p.slope);
                  }
                  const bf16x2 pk = {(__bf16)f0, (__bf16)f1};
                  d[e2] = __builtin_bit_cast(unsigned, pk);
                } else {
                  d[e2] = __builtin_amdgcn_perm(hi, lo, (jj & 1) ? 0x07060302u : 0x05040100u);
                }
              }
              *reinterpret_cast<u32x4*>(Xs + ((size_t)(g8 * XW + pos + jj)) * 16) = d;
            }
          }
        } else if (X4) {
          const int pos = pb * 256 + 4 * lane;
          if (pos < XW) {  // (XW is a multiple of 64: the four positions of a lane are inside or outside together)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              frag v;
#pragma unroll
              for (int e = 0; e < CPG; ++e) {
                float f = xr[t][(X4 ? 4 : 1) * e + (X4 ? jj : 0)];
                if (LEAKY) f = fmaxf(f, f * p.slope);
                EL::set(v, e, f);
              }
              *reinterpret_cast<frag*>(Xs + ((size_t)(g8 * XW + pos + jj)) * 16) = v;
            }
          }
        } else {
        frag v;
#pragma unroll
        for (int e = 0; e < CPG; ++e) {
          float f = xr[t][e];
          if (LEAKY) f = fmaxf(f, f * p.slope);  // slope in [0, 1)
          EL::set(v, e, f);
        }
        *reinterpret_cast<frag*>(Xs + ((size_t)(g8 * XW + pb * 64 + lane)) * 16) = v;
        }
      }
    }
  };

  const int ch_begin = (int)((long long)kz * tg.nch / tg.ks), ch_end = (int)((long long)(kz + 1) * tg.nch / tg.ks);
  if (NP && wave >= NW) {  // producer waves: one chunk ahead of the MFMA waves, one barrier per chunk like them
    __builtin_amdgcn_s_setprio(3);
    // (a chunk with more staging tasks than the producers' registers hold goes through them in batches)
    issueA(ch_begin, 0);
    for (int tb = 0; tb < ntask; tb += MAXT * NS) {
      loadX(ch_begin, tb);
      storeX(0, tb);
    }
    __syncthreads();
    for (int ch = ch_begin; ch < ch_end; ++ch) {
      const int cb = (ch - ch_begin) & 1;
      if (ch + 1 < ch_end) {
        issueA(ch + 1, cb ^ 1);
        for (int tb = 0; tb < ntask; tb += MAXT * NS) {
          loadX(ch + 1, tb);
          storeX(cb ^ 1, tb);
        }
      }
      __syncthreads();
    }
    return;
  }
  if (!NP) {
    issueA(ch_begin, 0);
    loadX(ch_begin);
    storeX(0);
  }
  __syncthreads();
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int cb = (ch - ch_begin) & 1;
    const bool more = !NP && ch + 1 < ch_end;
    if (more) {
      issueA(ch + 1, cb ^ 1);
      loadX(ch + 1);
    }
    const char* As = smem + cb * tg.buf_bytes;
    const char* Xs = As + tg.a_bytes;
    for (int cg = 0; cg < ncg; ++cg) {
      const char* Ab = As + ((size_t)(cg * 2 + h) * BM + wm * TM * 32 + l31) * 16;
      const char* Xb = Xs + (size_t)cg * 2 * XW * 16;
      for (int j = 0; j < K; ++j) {
        frag a[TM], bb[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
          a[tm] = *reinterpret_cast<const frag*>(Ab + ((size_t)j * ncg * 2 * BM + tm * 32) * 16);
        const int xo = j * p.dj * P * 16;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bb[tn] = *reinterpret_cast<const frag*>(Xb + laneoff[tn] + xo);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = EL::mma(a[tm], bb[tn], acc[tm][tn]);
      }
    }
    if (more) storeX(cb ^ 1);
    __syncthreads();  // publishes chunk ch+1 (LDS writes + the weight DMA) and retires the reads of chunk ch
  }

  conv_tile_epilogue<TM, TN, IO>(p, tg, acc, smem, part, wave, wm, wn, lane, b, kz, u0, m0, oo, BM);
}


struct Plan {
  int variant;
  int ppl;  // consecutive positions per lane and channel of the input loads: 1 (dword loads), 4 (16-byte loads of fp32:
            // kernel template X4) or 8 (16-byte loads of bf16 activations: IO & 1)
  int BM, BN, NW;
  BfGeom g;
  size_t scratch_floats, pack_bytes, lds_bytes;
};

constexpr int MAXT = 5;      // staging tasks (CPG loads each) a wave keeps in flight
constexpr int MAXT_WS = 10;  // ... a producer wave of a warp-specialised launch (it holds no accumulators)
constexpr int MAXT_X4 = 2;   // ... of 4 x CPG loaded floats each, with 16-byte loads
constexpr int MAXT_X4_WS = 4;

bool eligible(const VcvConvArgs& a, int io = 0) {
  if (a.io != io || (io == 0 && a.post_scale != 0.f)) return false;
  if (io != 0 && (a.out_tf != VCV_TF_NONE || (((long long)a.Tin * a.P) & 1) || (((long long)a.Tout * a.P) & 1) || a.xaux || a.oaux))
    return false;  // bf16 tensors: rows of an even number of elements (4-byte aligned), no derivative masks
  const bool fwd_type = a.a_mode == 0 && a.phases <= 1 && a.ms <= 1;
  const bool phased = a.a_mode == 1 && a.phases > 1 && a.s == 1 && a.dj == -1 && a.ms <= 1;
  // all output phases of a transposed conv as rows of one launch (16-bit activations only: the epilogue that interleaves them)
  const bool merged = io != 0 && a.ms > 1 && a.a_mode == 1 && a.phases <= 1 && a.s == 1 && a.dj == -1 && a.os == a.ms && a.P == 1 &&
                      a.Mg % a.ms == 0 && !a.res && !a.mask && !a.accumulate;
  return (fwd_type || phased || merged) && a.G == 1 &&
         (a.in_tf == VCV_TF_NONE || (a.in_tf == VCV_TF_LEAKY && a.slope < 1.f && a.slope >= 0.f)) && a.Mg >= 32 &&
         a.Cg >= 16 && a.K <= 16 && a.s >= 1 && a.s <= 3 && (long long)a.Tin * a.P * 4 < (1ll << 31) &&
         (long long)a.Mg * a.Tout * a.P < (1ll << 31);
}

template <class EL>
bool make_plan(const VcvConvArgs& a, int BM, int BN, int NW, Plan& pl, int NS = 0, int ppl = 1) {
  const bool ws = NS != 0;  // warp-specialised variant: NS producer waves, which stage a chunk in as many batches as it takes
  if (NS == 0) NS = NW;
  constexpr int KG = 2 * EL::CPG, ESZ = EL::ESZ;  // channels per (h = 0, h = 1) group pair; bytes per element
  pl.BM = BM; pl.BN = BN; pl.NW = NW; pl.ppl = ppl;
  const bool x4 = ppl > 1;
  BfGeom& g = pl.g;
  const int qspan = (BN - 1) / a.P + 1;
  const int adj = a.dj < 0 ? -a.dj : a.dj;
  g.phases = a.phases > 1 ? a.phases : 1;
  g.JA = vcv_cdiv(a.K, g.phases);
  const int rowmax = (qspan * a.s + (g.JA - 1) * adj + 1) * a.P;
  g.xw = (rowmax + (ppl - 1) + 63) & ~63;  // (up to ppl - 1 elements of round-down at the start)
  // chunk depth: 16-channel groups per chunk.  Candidates must fit two LDS buffers (one when a single chunk covers the
  // reduction) and MAXT staging tasks per wave; among them the least zero-padded channel count wins, then the deeper.
  const int cmax = ((a.Cg + KG - 1) / KG) * KG;
  // LDS budget: narrow layers (few chunks per workgroup, so nothing inside a workgroup overlaps the staging latency)
  // run two workgroups per CU when a chunk depth fits 78 KiB; measured +40-60 % on the 64-channel generator layers and
  // the first period-discriminator convs, nothing on the deep layers, which take the whole LDS for deeper chunks
  const bool narrow = a.Cg <= 64 || (a.Cg <= 128 && a.s > 1);
  int bkc = 0;
  for (int pass = narrow ? 0 : 1; pass < 2 && bkc == 0; ++pass) {
    const size_t lds_cap = (pass == 0 ? 78 : 156) * 1024;
    long long best_pad = 1ll << 60;
    for (int cand = 64; cand >= KG; cand -= KG) {
      if (cand > cmax) continue;
      const int nch = vcv_cdiv(a.Cg, cand);
      const size_t buf = (size_t)g.JA * cand * BM * ESZ + (size_t)cand * g.xw * ESZ;
      if ((nch > 1 ? 2 : 1) * buf > lds_cap || (!ws && (cand / EL::CPG) * (x4 ? (g.xw + 64 * ppl - 1) / (64 * ppl) : g.xw >> 6) > (x4 ? MAXT_X4 : MAXT) * NS)) continue;
      const long long padded = (long long)nch * cand;
      if (padded < best_pad) best_pad = padded, bkc = cand;
    }
  }
  if (bkc == 0) return false;
  g.BKC = bkc;
  g.ncg = bkc / KG;
  g.nch = vcv_cdiv(a.Cg, bkc);
  g.ntu = vcv_cdiv(a.Q * a.P, BN);
  g.nmt = vcv_cdiv(a.Mg, BM);
  g.a_bytes = g.JA * bkc * BM * ESZ;
  g.buf_bytes = g.a_bytes + bkc * g.xw * ESZ;
  pl.lds_bytes = (g.nch > 1 ? 2ull : 1ull) * g.buf_bytes;  // one chunk: no second buffer, more workgroups per CU
  if (pl.lds_bytes > VCV_LDS_LIMIT) return false;
  g.ks = 1;
  g.vec = 0;
  const int xcd_remap = vcv_tuning().xcd_remap;
  // nothing to share when a column tile has one workgroup (measured: the re-deal alone costs the 64 x 10 s decode 11 %:
  // eight XCDs walking eight far-apart regions of the tensor instead of one)
  g.xcd = xcd_remap && g.nmt * (g.phases > 1 ? g.phases : 1) > 1;
  pl.pack_bytes = (size_t)g.phases * g.nmt * g.nch * g.a_bytes;
  pl.scratch_floats = 0;
  return true;
}

// variants: 0: 128x256 / 8 waves (2x2 per wave)   1: 128x128 / 8 waves (2x1)   2: 128x224 / 14 waves (2x1)
//           3: 64x256 / 8 waves (1x2... 2x4 waves of 1x2)   4: 64x128 / 8 waves (1x1)   5: 32x256 / 8 waves (1x1)
//           6: 64x224 / 14 waves (1x1)   7: 128x288 / 9 waves (4x1)   16: 64x288 / 9 waves (2x1)
//           12 / 13: warp-specialised 128x256 (8 + 4 waves) / 128x224 (4 + 4 waves)
template <class EL, int IO = 0>
bool choose(const VcvConvArgs& a, Plan& pl) {
  const int U = a.Q * a.P;
  if (U < 96) return false;
  const int nph = a.phases > 1 ? a.phases : 1;
  auto blocks = [&](int bm, int bn) { return (long long)a.B * vcv_cdiv(U, bn) * vcv_cdiv(a.Mg, bm) * nph; };
  bool ok = false;
  // fill of the last round of 256 workgroups x useful columns of the position tiles
  auto eff2 = [&](int bm, int bn) {
    const long long nb = blocks(bm, bn);
    const long long rounds = (nb + 255) / 256;
    return ((double)U / ((double)vcv_cdiv(U, bn) * bn)) * ((double)nb / (double)(rounds * 256));
  };
  if (a.Mg >= 128) {
    // (the 3-phase data gradients of the 512-channel period layers: 384 tiles of 128 rows run 1.5 rounds, 768 of 64 rows 3)
    if (U > 160 && U <= 224 && eff2(64, 224) > eff2(128, 224) + 0.2 && make_plan<EL>(a, 64, 224, 14, pl)) pl.variant = 6, ok = true;
    else if (U > 160 && U <= 224 && make_plan<EL>(a, 128, 224, 14, pl)) pl.variant = 2, ok = true;
    else if (U > 256 && U <= 288 && eff2(64, 288) > eff2(128, 288) + 0.2 && make_plan<EL>(a, 64, 288, 9, pl)) pl.variant = 16, ok = true;
    else if (U > 256 && U <= 288 && make_plan<EL>(a, 128, 288, 9, pl)) pl.variant = 7, ok = true;
    else {
      // tile width by efficiency = (useful columns of the position tiles) x (fill of the last round of 256 workgroups):
      // the 128 -> 512 period layers have 608-629 positions per batch element: 3 tiles of 256 waste a fifth of the
      // columns and 384 workgroups run 1.5 rounds (0.59), 2 tiles of 320 waste 3-5 % in exactly one round (0.95)
      auto eff = [&](int bn) {
        const long long nb = blocks(128, bn);
        const long long rounds = (nb + 255) / 256;
        return ((double)U / ((double)vcv_cdiv(U, bn) * bn)) * ((double)nb / (double)(rounds * 256));
      };
      const double e256 = (U > 160 && blocks(128, 256) >= 256) ? eff(256) : 0.0, e128 = eff(128);
      const double e320 = (U > 320 && blocks(128, 320) >= 160) ? eff(320) : 0.0;
      if (e320 > e256 + 0.08 && e320 > e128 + 0.08 && make_plan<EL>(a, 128, 320, 8, pl)) pl.variant = 11, ok = true;
      else if (e256 > 0.0 && make_plan<EL>(a, 128, 256, 8, pl)) pl.variant = 0, ok = true;
      else if (make_plan<EL>(a, 128, 128, 8, pl)) pl.variant = 1, ok = true;
    }
  } else if (a.Mg >= 64) {
    if (U > 160 && U <= 224 && make_plan<EL>(a, 64, 224, 14, pl)) pl.variant = 6, ok = true;
    else if (U > 160 && blocks(64, 256) >= 256 && make_plan<EL>(a, 64, 256, 8, pl)) pl.variant = 3, ok = true;
    else if (make_plan<EL>(a, 64, 128, 8, pl)) pl.variant = 4, ok = true;
  } else {
    if (U >= 2048 && blocks(32, 512) >= 256 && make_plan<EL>(a, 32, 512, 8, pl)) pl.variant = 8, ok = true;
    else if (make_plan<EL>(a, 32, 256, 8, pl)) pl.variant = 5, ok = true;
  }
  // warp-specialised twins of the wide fp32 tiles (8 or 4 MFMA waves + 4 producer waves): measured +5 % on the
  // 1024-channel period layers, +10 % on the 128-channel generator layers -- where 128 workgroups of 128x256 with
  // producers beat 256 of the plain 128x128 tile; the phased data gradients (two taps per staged span: staging-bound),
  // the stride-3 layers (three times the span per position: 105 -> 87 TFLOP/s with four producers) and the
  // 128x320 / 128x128 twins measured slower and keep every wave staging
  const bool no_ws = !vcv_tuning().pk_ws;
  const bool ws_bf16 = vcv_tuning().pk_ws_bf16 != 0;  // (experiment switch)
  if (ok && !no_ws && a.Mg >= 128 && (EL::ESZ == 4 || ws_bf16) && nph == 1 && a.s == 1) {
    Plan p2;
    if (pl.variant == 0 && make_plan<EL>(a, 128, 256, 8, p2, 4)) pl = p2, pl.variant = 12;
    else if (pl.variant == 2 && make_plan<EL>(a, 128, 224, 4, p2, 4)) pl = p2, pl.variant = 13;
    else if (pl.variant == 1 && U > 160 && blocks(128, 256) >= 112 && make_plan<EL>(a, 128, 256, 8, p2, 4)) pl = p2, pl.variant = 12;
  }
  // 16-byte input loads (kernel template X4): a quarter of the load instructions of the staging phase.  Measured
  // against dword loads: the stride-3 layers +4-8 % forward and +8-11 % in their phased data gradients, the rest
  // +0-3 %, the bf16 conv class 264 -> 287 TFLOP/s in the step; only the 32-row phased data gradients (128 -> 32
  // channels: 39 -> 36) and the bf16 launches with <= 64 input channels (278 -> 252 on the 64-channel k7 layers: two
  // workgroups per CU there, and the 32 staged floats per task cost registers) lose and keep dword loads
  const bool no_x4 = !vcv_tuning().pk_x4;
  if (ok && (IO & 1)) {
    // bf16 activations: always 16-byte loads of eight positions (the only loader of a bf16 `x`)
    Plan p2;
    if (pl.variant == 12 || pl.variant == 13 || !make_plan<EL>(a, pl.BM, pl.BN, pl.NW, p2, 0, 8)) return false;
    p2.variant = pl.variant;
    pl = p2;
  } else if (ok && !no_x4 && !(nph > 1 && a.Mg < 64) && !(EL::ESZ == 2 && a.Cg <= 64)) {
    Plan p2;
    const int ns = pl.variant == 12 || pl.variant == 13 ? 4 : 0;
    if (make_plan<EL>(a, pl.BM, pl.BN, pl.NW, p2, ns, 4)) { p2.variant = pl.variant; pl = p2; }
  }
  if (!ok) return false;
  // too few tiles for 256 CUs: split the reduction over ks blocks per tile (deterministic slabs + finishing pass),
  // aiming at one full round of resident workgroups (256 x the workgroups a CU holds at this LDS footprint)
  const long long nb = blocks(pl.BM, pl.BN);
  if (IO == 0 && nph == 1 && nb < 192 && pl.g.nch >= 4) {  // (bf16 activations: no split -- the finishing passes are fp32-only)
    // (bf16 chunks are short enough that the finishing pass outweighs a second resident round: measured, it keeps
    // the flat 384 target)
    const long long target = pl.lds_bytes * 2 <= VCV_LDS_LIMIT ? 512 : 256;
    long long ks = EL::ESZ == 4 ? (target + nb / 2) / nb : (384 + nb - 1) / nb;
    if (ks > pl.g.nch / 2) ks = pl.g.nch / 2;
    if (ks >= 2) {
      pl.g.ks = (int)ks;
      pl.scratch_floats = (size_t)ks * a.B * a.Mg * U;
    }
  }
  // 16-byte epilogue through LDS: output rows contiguous in the column index, room for a 32 x 40 float tile per MFMA wave
  const bool no_vec = !vcv_tuning().pk_vec;
  pl.g.vec = (!no_vec && nph == 1 && a.os == 1 && a.oo == 0 && (!a.mask || a.P == 1) &&
              (size_t)pl.NW * 32 * 40 * 4 <= pl.lds_bytes) ? 1 : 0;
  // bf16 `y`: the 16-byte stores carry eight columns: rows of a multiple of eight elements, 16-byte aligned tensors
  if ((IO & 2) && ((a.Tout * a.P) % 8 != 0 || (((uintptr_t)a.y | (uintptr_t)a.res) & 15))) pl.g.vec = 0;
  if (a.ms > 1) pl.g.vec = 0;  // (merged phases: the interleaving epilogue)
  return true;
}

template <class EL, int TM, int TN, int WM, int WN, int NP = 0, int IO = 0>
int launch(const VcvConvArgs& a, const Plan& pl, typename EL::frag* wp, float* part, int flip, bool pack_valid, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, NT = 64 * (WM * WN + NP);
  const BfGeom& g = pl.g;
  if (!pack_valid) {
    const size_t total = pl.pack_bytes / 16;
    const int mode = a.ms > 1 ? 3 : (g.phases > 1 ? 2 : (flip ? 1 : 0));
    hipLaunchKernelGGL(pack_pk_kernel<EL>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.w, wp, a.Mg, a.Cg, a.K,
                       BM, g.BKC, g.JA, g.nch, g.nmt, a.ms > 1 ? a.ms : g.phases, mode, total);
  }
  void (*kern)(const VcvConvArgs, const BfGeom, const typename EL::frag*, float*);
  if constexpr (IO != 0) {
    static_assert((IO & 3) == 3 && NP == 0, "16-bit activations: x, y and res together; no producer-wave variants");
    if (pl.ppl != 8) return VCV_EINVAL;
    kern = a.in_tf == VCV_TF_LEAKY ? conv_pk_kernel<EL, TM, TN, WM, WN, true, MAXT_X4, 0, true, IO>
                                   : conv_pk_kernel<EL, TM, TN, WM, WN, false, MAXT_X4, 0, true, IO>;
  } else {
    kern = pl.ppl == 4 ? (a.in_tf == VCV_TF_LEAKY ? conv_pk_kernel<EL, TM, TN, WM, WN, true, (NP ? MAXT_X4_WS : MAXT_X4), NP, true>
                                                  : conv_pk_kernel<EL, TM, TN, WM, WN, false, (NP ? MAXT_X4_WS : MAXT_X4), NP, true>)
                       : (a.in_tf == VCV_TF_LEAKY ? conv_pk_kernel<EL, TM, TN, WM, WN, true, (NP ? MAXT_WS : MAXT), NP, false>
                                                  : conv_pk_kernel<EL, TM, TN, WM, WN, false, (NP ? MAXT_WS : MAXT), NP, false>);
  }
  if (pl.lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds_bytes) != hipSuccess)
    return VCV_EHIP;
  dim3 grid(a.B * g.ntu * g.ks, g.nmt, g.phases), block(NT);
  const double flops = 2.0 * a.B * a.Mg * a.Cg * a.K * a.P * (double)(g.phases > 1 ? a.Tin : a.Q);
  const int tag[12] = {a.B, EL::ESZ == 2 ? 2 : 4, a.Cg, a.Mg, a.K, a.Q, a.P, a.s, g.phases, a.a_mode + 10 * g.ks, BM * 1000 + pl.BN, g.BKC};
  const double esz = IO ? 2.0 : 4.0;  // bytes per activation element in HBM
  const double abytes = esz * (double)a.B * a.Cg * a.Tin * a.P + 4.0 * (double)a.Mg * a.Cg * a.K +
                        esz * (double)a.B * (a.ms > 1 ? a.Mg / a.ms : a.Mg) * a.Tout * a.P * (1 + (a.res ? 1 : 0) + (a.oaux ? 1 : 0) + (a.accumulate ? 1 : 0));
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_CONV_DMA, flops, tag, 12, &ev0, &ev1, abytes, EL::ESZ == 2 ? flops / VCV_PEAK_BF16_MFMA : 0.0);
  VCV_LAUNCH_EV(kern, grid, block, (unsigned)pl.lds_bytes, st, ev0, ev1, a, g, (const typename EL::frag*)wp, part);
  if (g.ks > 1) {
    const size_t n = (size_t)a.B * a.Mg * a.Q * a.P;
    if (g.vec && !a.mask && n % 4 == 0 && a.Q == a.Tout && a.Q * a.P >= 4)
      hipLaunchKernelGGL(conv_pk_finish4_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, a, (const float*)part, g.ks);
    else
      hipLaunchKernelGGL(conv_pk_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, (const float*)part, g.ks);
  }
  return vcv_check_launch();
}

// fp32: with 16-byte input loads this kernel is ahead of the LDS-DMA kernel (conv_dma.hip) on every shape of the step
// it was behind on before (32-channel layers +6-9 %, wide k <= 3 layers +12-38 %); conv_dma.hip keeps 16..31 channels
template <class EL>
bool wanted(const VcvConvArgs& a) {
  if (EL::ESZ == 2) return true;
  return a.Cg >= 32;
}

template <class EL, int IO = 0>
int plan_t(const VcvConvArgs* args, int flip, int64_t* out) {
  if (!args || !out || !eligible(*args, IO) || !wanted<EL>(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose<EL, IO>(*args, pl)) return VCV_EINVAL;
  out[0] = (int64_t)((pl.pack_bytes + 3) / 4);
  out[1] = (int64_t)pl.scratch_floats;
  const BfGeom& g = pl.g;
  out[2] = ((int64_t)(EL::ESZ == 2 ? 2 : 1) << 61) | ((int64_t)pl.BM << 40) | ((int64_t)g.BKC << 28) | ((int64_t)g.JA << 20) |
           ((int64_t)g.phases << 8) | ((int64_t)(args->ms > 1 ? args->ms : 0) << 1) | (flip ? 1 : 0);
  return 0;
}

template <class EL, int IO = 0>
int run_t(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid, void* stream) {
  if (!args || !pack_ws || !eligible(*args, IO) || !wanted<EL>(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose<EL, IO>(*args, pl)) return VCV_EINVAL;
  if (pl.g.ks > 1 && !scratch_ws) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  typename EL::frag* wp = reinterpret_cast<typename EL::frag*>(pack_ws);
  const bool pv = pack_valid != 0;
  if constexpr (IO != 0) {
    switch (pl.variant) {
      case 0: return launch<EL, 2, 2, 2, 4, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 1: return launch<EL, 2, 1, 2, 4, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 2: return launch<EL, 2, 1, 2, 7, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 3: return launch<EL, 1, 2, 2, 4, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 4: return launch<EL, 1, 1, 2, 4, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 5: return launch<EL, 1, 1, 1, 8, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 6: return launch<EL, 1, 1, 2, 7, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 8: return launch<EL, 1, 2, 1, 8, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 16: return launch<EL, 2, 1, 1, 9, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 11: return launch<EL, 1, 5, 4, 2, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      case 7: return launch<EL, 4, 1, 1, 9, 0, IO>(*args, pl, wp, scratch_ws, flip, pv, st);
      default: return VCV_EINVAL;
    }
  } else {
  switch (pl.variant) {
    case 0: return launch<EL, 2, 2, 2, 4>(*args, pl, wp, scratch_ws, flip, pv, st);
    case 1: return launch<EL, 2, 1, 2, 4>(*args, pl, wp, scratch_ws, flip, pv, st);
    case 2: return launch<EL, 2, 1, 2, 7>(*args, pl, wp, scratch_ws, flip, pv, st);
    case 3: return launch<EL, 1, 2, 2, 4>(*args, pl, wp, scratch_ws, flip, pv, st);
    case 4: return launch<EL, 1, 1, 2, 4>(*args, pl, wp, scratch_ws, flip, pv, st);
    case 5: return launch<EL, 1, 1, 1, 8>(*args, pl, wp, scratch_ws, flip, pv, st);
    case 6: return launch<EL, 1, 1, 2, 7>(*args, pl, wp, scratch_ws, flip, pv, st);
    case 8: return launch<EL, 1, 2, 1, 8>(*args, pl, wp, scratch_ws, flip, pv, st);   // 32 x 512
    case 16: return launch<EL, 2, 1, 1, 9>(*args, pl, wp, scratch_ws, flip, pv, st);  // 64 x 288, 9 waves of 64 rows x 32 columns
    case 12: return launch<EL, 2, 2, 2, 4, 4>(*args, pl, wp, scratch_ws, flip, pv, st);  // 128 x 256, 8 MFMA + 4 producer waves
    case 13: return launch<EL, 1, 7, 4, 1, 4>(*args, pl, wp, scratch_ws, flip, pv, st);  // 128 x 224, 4 MFMA waves of 32 x 224 + 4 producers
    case 11: return launch<EL, 1, 5, 4, 2>(*args, pl, wp, scratch_ws, flip, pv, st);  // 128 x 320: 8 waves of 32 rows x 5 column tiles
    default: return launch<EL, 4, 1, 1, 9>(*args, pl, wp, scratch_ws, flip, pv, st);
  }
  }
}

template <class EL>
int pack_job_t(const VcvConvArgs* args, int flip, VcvPackJob* out) {
  if (!args || !out || !eligible(*args) || !wanted<EL>(*args)) return VCV_EINVAL;
  Plan pl;
  if (!choose<EL>(*args, pl)) return VCV_EINVAL;
  const BfGeom& g = pl.g;
  out->kind = EL::ESZ == 2 ? 2 : 1;
  out->M = args->Mg, out->C = args->Cg, out->K = args->K;
  out->BM = pl.BM, out->BKC = g.BKC, out->JA = g.JA, out->nch = g.nch, out->nmt = g.nmt, out->phases = g.phases;
  out->mode = g.phases > 1 ? 2 : (flip ? 1 : 0);
  out->total = (int64_t)(pl.pack_bytes / 16);
  return VCV_OK;
}

}  // namespace

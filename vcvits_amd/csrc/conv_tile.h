// conv_tile.h -- tile geometry, output epilogue and split-reduction finishing passes shared by the packed-operand
// implicit-GEMM kernels (conv_pk.hip: fp32 / bf16 elements; conv_x3.hip: fp32 operands as three bf16 terms).
#pragma once
#include "common.h"

namespace {

struct BfGeom {
  int BKC;      // reduction channels per chunk (multiple of 2 * CPG: 16 for bf16, 8 for fp32)
  int ncg;      // BKC / (2 * CPG)
  int nch;      // chunks
  int ntu;      // position tiles per batch element
  int nmt;      // M tiles
  int xw;       // staged span (positions, a multiple of 64)
  int a_bytes;  // JA * BKC * BM * 2
  int buf_bytes;  // a_bytes + BKC * xw * 2
  int JA;         // taps stored per channel in a weight slab (K, or ceil(K/phases) for a phased launch)
  int phases;     // > 1: transposed / strided-data-gradient launch, one residue per blockIdx.z
  int ks;         // > 1: the chunks are split over ks blocks per tile, partial sums go to a scratch slab each
  int xcd;        // 1: XCD-aware tile order (xcd_tile_id below)
  int vec;        // 1: rows of the output are contiguous in the column index and the LDS has room for a 32 x 40 tile per
                  // wave: the epilogue goes through LDS and stores 16 bytes per lane (4 consecutive columns of one row)
};

// Epilogue of one workgroup tile: acc[tm][tn] are the 32 x 32 MFMA accumulators of this wave (rows (wm*TM + tm)*32 ..,
// columns (wn*TN + tn)*32 ..), `smem` the workgroup's LDS (free: the caller's main loop ended with a barrier), `part`
// the partial-sum slabs of a split launch.  b: batch element, kz: split index, u0 / m0: first column / row of the tile,
// oo: output row offset (phase residue included).
// Tile of this workgroup, XCD-aware.  The grid is (column tiles x splits, m-tiles, phases); all m-tiles (and phases) of one
// column tile read the SAME input span, but in dispatch order they are a whole grid row apart, and workgroups are dealt to the
// 8 XCDs round robin (MI355X_MICROARCH.md: blocks b and b + 8 share an XCD, each XCD has its own L2): the span was fetched
// into up to 8 L2s (round 3: 2.2x the algorithmic HBM bytes on the dominant class).  Here the linear workgroup id is
// re-dealt so that consecutive ids sit on ONE XCD (the bijective form of cdna_hip_programming.md T1) and the m-tile /
// phase index runs fastest: the workgroups sharing a span are neighbours in time on one L2.  A speed-only mapping: any
// placement is correct.  VCVITS_NO_XCD_REMAP=1 (read once by the planners: BfGeom.xcd) restores the plain order.
__device__ __forceinline__ void xcd_tile_id(int& bxk, int& mt, int& r, int remap) {
  if (!remap) {
    bxk = blockIdx.x, mt = blockIdx.y, r = blockIdx.z;
    return;
  }
  const unsigned nmt = gridDim.y, nph = gridDim.z;
  const unsigned n = gridDim.x * nmt * nph;
  const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + nmt * blockIdx.z);  // dispatch order: x fastest
  const unsigned q = n >> 3, rem = n & 7, xcd = id & 7;
  unsigned lid = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + (id >> 3);
  mt = (int)(lid % nmt);
  lid /= nmt;
  r = (int)(lid % nph);
  bxk = (int)(lid / nph);
}

// Storage kinds of an activation tensor in HBM: 0 = fp32, 1 = bf16, 2 = fp16 (IEEE half: the residual stream of the decoder
// in bf16-activation mode -- 11 significand bits, so re-rounding the stream at every residual add costs 1/64 of the error
// energy a bf16 stream would; values are clamped to the finite fp16 range before the conversion)
__device__ __forceinline__ float us_to_f32(unsigned short u, int kind) {
  return kind == 2 ? (float)__builtin_bit_cast(_Float16, u) : __uint_as_float((unsigned)u << 16);
}
template <int KIND>
__device__ __forceinline__ unsigned short f32_to_us(float v) {
  if constexpr (KIND == 2) return __builtin_bit_cast(unsigned short, (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f));
  else return __builtin_bit_cast(unsigned short, (__bf16)v);  // round to nearest even (v_cvt_pk_bf16_f32)
}
template <int KIND>
__device__ __forceinline__ float ld_act(const float* p, size_t i) {
  if constexpr (KIND != 0) return us_to_f32(reinterpret_cast<const unsigned short*>(p)[i], KIND);
  else return p[i];
}
template <int KIND>
__device__ __forceinline__ void st_act(float* p, size_t i, float v) {
  if constexpr (KIND != 0) reinterpret_cast<unsigned short*>(p)[i] = f32_to_us<KIND>(v);
  else p[i] = v;
}

// IO bit 1 (value 2): `y`, `res` and the accumulate target are 16-bit tensors (conv_pk_io*.hip) -- bf16, or fp16 with bit 3
// (value 8); the split-reduction paths are fp32-only (the planner never splits such a launch).
template <int TM, int TN, int IO = 0>
__device__ __forceinline__ void conv_tile_epilogue(const VcvConvArgs& p, const BfGeom& tg, f32x16 (&acc)[TM][TN], char* smem,
                                                   float* __restrict__ part, int wave, int wm, int wn, int lane, int b, int kz,
                                                   int u0, int m0, int oo, int BM) {
  const int l31 = lane & 31, h = lane >> 5;
  const int P = p.P, U = p.Q * P, Mg = p.Mg;
  const int rows_valid = Mg - m0 < BM ? Mg - m0 : BM;
  const bool mtail = m0 + BM > Mg;
  constexpr bool YB = (IO & 2) != 0;
  constexpr int YK = YB ? ((IO & 8) ? 2 : 1) : 0;  // storage kind of y / res
  if constexpr (YB) {
    if (TM * TN <= 4 && tg.vec) {
      // ---- 16-byte epilogue, bf16 output: as below, each lane taking EIGHT consecutive columns of one row (two 16-byte
      // LDS reads, one 16-byte store; `res` / the accumulate target are read the same way)
      typedef unsigned short us8 __attribute__((ext_vector_type(8)));
      float* T = reinterpret_cast<float*>(smem) + wave * (32 * 40);
      const unsigned rowstride = (unsigned)(p.Tout * P);
      const size_t ybase = ((size_t)b * Mg + m0) * rowstride;
      const float* bias = p.bias ? p.bias + m0 : nullptr;
      unsigned short* y16 = reinterpret_cast<unsigned short*>(p.y);
      const unsigned short* r16 = reinterpret_cast<const unsigned short*>(p.res);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
          for (int e = 0; e < 16; ++e) T[((e & 3) + 8 * (e >> 2) + 4 * h) * 40 + l31] = acc[tm][tn][e];
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int r = ps * 16 + (lane >> 2), c8 = lane & 3;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(T + r * 40 + 8 * c8);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(T + r * 40 + 8 * c8 + 4);
            const int ml = (wm * TM + tm) * 32 + r;
            const int u = u0 + (wn * TN + tn) * 32 + 8 * c8;
            if ((mtail && ml >= rows_valid) || u >= U) continue;
            const size_t idx = ybase + (size_t)((unsigned)ml * rowstride) + u;
            const int nv = U - u < 8 ? U - u : 8;
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            float rr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, yy[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            float mk[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
            if (p.mask) {  // (P == 1: the mask row of this batch element is indexed by u)
              const float* mrow = p.mask + (size_t)b * p.Tout + u;
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (j < nv) mk[j] = mrow[j];
            }
            if (nv == 8) {
              if (p.res) {
                const us8 t8 = *reinterpret_cast<const us8*>(r16 + idx);
#pragma unroll
                for (int j = 0; j < 8; ++j) rr[j] = us_to_f32(t8[j], YK);
              }
              if (p.accumulate) {
                const us8 t8 = *reinterpret_cast<const us8*>(y16 + idx);
#pragma unroll
                for (int j = 0; j < 8; ++j) yy[j] = us_to_f32(t8[j], YK);
              }
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                if (j < nv) {
                  if (p.res) rr[j] = ld_act<YK>(p.res, idx + j);
                  if (p.accumulate) yy[j] = ld_act<YK>(p.y, idx + j);
                }
              }
            }
            const float bv = bias ? bias[ml] : 0.f;
            const float ps_ = p.post_scale != 0.f ? p.post_scale : 1.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              float x = p.alpha * v[j] + bv;
              x = vcv_act(x, p.out_act, p.slope);
              x += rr[j];
              x *= mk[j];
              x = x * ps_ + yy[j];
              v[j] = x;
            }
            if (nv == 8) {
              us8 o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o[j] = f32_to_us<YK>(v[j]);
              *reinterpret_cast<us8*>(y16 + idx) = o;
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (j < nv) st_act<YK>(p.y, idx + j, v[j]);
            }
          }
        }
      }
      return;
    }
  }
  if (!YB && TM * TN <= 4 && tg.vec) {  // (compile-time bound: the unrolled body of the 5- and 7-tile waves would not stay in registers)
    // ---- 16-byte epilogue: each 32 x 32 accumulator tile goes through a wave-private LDS tile (pitch 40 floats: the
    // two row halves of the MFMA layout land 32 banks apart), comes back as rows of four consecutive columns per lane,
    // and the epilogue operands (residual, activation-derivative mask, accumulate) are read the same way: a quarter
    // of the global memory instructions of the dword-per-lane path below.  Needs os == 1, oo == 0, i.e. output index
    // = row * rowstride + u, and a row mask only with P == 1 (mask index = u; the launcher sets tg.vec).  The LDS is free: the chunk loop ended
    // with a barrier, producers of a specialised launch have left.
    float* T = reinterpret_cast<float*>(smem) + wave * (32 * 40);
    const unsigned rowstride = (unsigned)(p.Tout * P);
    const bool split = tg.ks > 1;
    const size_t ybase = split ? (((size_t)kz * p.B + b) * Mg + m0) * (size_t)U : ((size_t)b * Mg + m0) * rowstride;
    const unsigned rs = split ? (unsigned)U : rowstride;
    float* __restrict__ yout = split ? part : p.y;
    const float* bias = (!split && p.bias) ? p.bias + m0 : nullptr;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
        for (int e = 0; e < 16; ++e) T[((e & 3) + 8 * (e >> 2) + 4 * h) * 40 + l31] = acc[tm][tn][e];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int r = ps * 8 + (lane >> 3), c4 = lane & 7;
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(T + r * 40 + 4 * c4);
          const int ml = (wm * TM + tm) * 32 + r;
          const int u = u0 + (wn * TN + tn) * 32 + 4 * c4;
          if ((mtail && ml >= rows_valid) || u >= U) continue;
          const size_t idx = ybase + (size_t)((unsigned)ml * rs) + u;
          const int nv = U - u < 4 ? U - u : 4;
          float v[4] = {a4[0], a4[1], a4[2], a4[3]};
          if (split) {
            if (nv == 4) *reinterpret_cast<f32x4*>(yout + idx) = a4;
            else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (j < nv) yout[idx + j] = v[j];  // (constant indices: a runtime trip count would put v[] in scratch)
            }
            continue;
          }
          float oa[4] = {0.f, 0.f, 0.f, 0.f}, rr[4] = {0.f, 0.f, 0.f, 0.f}, yy[4] = {0.f, 0.f, 0.f, 0.f};
          float mk[4] = {1.f, 1.f, 1.f, 1.f};
          if (p.mask) {  // (P == 1: the mask row of this batch element is indexed by u)
            const float* mrow = p.mask + (size_t)b * p.Tout + u;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nv) mk[j] = mrow[j];
          }
          if (nv == 4) {
            if (p.out_tf >= VCV_TF_DLEAKY) { const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.oaux + idx); oa[0] = t4[0], oa[1] = t4[1], oa[2] = t4[2], oa[3] = t4[3]; }
            if (p.res) { const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.res + idx); rr[0] = t4[0], rr[1] = t4[1], rr[2] = t4[2], rr[3] = t4[3]; }
            if (p.accumulate) { const f32x4 t4 = *reinterpret_cast<const f32x4*>(p.y + idx); yy[0] = t4[0], yy[1] = t4[1], yy[2] = t4[2], yy[3] = t4[3]; }
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (j < nv) {
                if (p.out_tf >= VCV_TF_DLEAKY) oa[j] = p.oaux[idx + j];
                if (p.res) rr[j] = p.res[idx + j];
                if (p.accumulate) yy[j] = p.y[idx + j];
              }
            }
          }
          const float bv = bias ? bias[ml] : 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = p.alpha * v[j] + bv;
            x = vcv_act(x, p.out_act, p.slope);
            if (p.out_tf == VCV_TF_DLEAKY) x *= vcv_dleaky(oa[j], p.slope);
            else if (p.out_tf == VCV_TF_DRELU) x = oa[j] > 0.f ? x : 0.f;
            else if (p.out_tf == VCV_TF_DTANH) x *= 1.f - oa[j] * oa[j];
            x += rr[j];
            x *= mk[j];
            x += yy[j];
            v[j] = x;
          }
          if (nv == 4) *reinterpret_cast<f32x4*>(p.y + idx) = f32x4{v[0], v[1], v[2], v[3]};
          else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (j < nv) p.y[idx + j] = v[j];
          }
        }
      }
    }
    return;
  }
  if (tg.ks > 1) {
    float* pb = part + (((size_t)kz * p.B + b) * Mg + m0) * (size_t)U;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
      const int u = u0 + (wn * TN + tn) * 32 + l31;
      if (u >= U) continue;
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int ml = (wm * TM + tm) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (mtail && ml >= rows_valid) continue;
          pb[(size_t)ml * U + u] = acc[tm][tn][e];
        }
    }
    return;
  }
  if constexpr (YB) {
    if (p.ms > 1) {
      // ---- merged phases of a transposed conv: row = (cout, phase), column = q; element -> y[b, cout, q * ms + phase + oo].
      // The four accumulator registers e & 3 = 0..3 of a lane are four consecutive rows: with ms a multiple of four they are
      // four consecutive phases of one channel, i.e. four consecutive output samples -- one 8-byte store when aligned, and the
      // 32 columns of a half-wave then cover 32 * ms contiguous samples.
      const int ms = p.ms, M = Mg / ms;
      unsigned short* y16 = reinterpret_cast<unsigned short*>(p.y);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int q = u0 + (wn * TN + tn) * 32 + l31;
        if (q >= U) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const int row0 = m0 + (wm * TM + tm) * 32 + 8 * e4 + 4 * h;  // rows row0 .. row0 + 3 (registers 4 e4 .. 4 e4 + 3)
            float v[4];
            int trow[4], co[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int row = row0 + j;
              co[j] = row / ms;
              trow[j] = q * ms + (row - co[j] * ms) + oo;
              float x = p.alpha * acc[tm][tn][4 * e4 + j];
              if (p.bias && row < Mg) x += p.bias[co[j]];
              v[j] = vcv_act(x, p.out_act, p.slope);
            }
            const size_t idx0 = ((size_t)b * M + co[0]) * (size_t)p.Tout + trow[0];
            if ((ms & 3) == 0 && row0 + 3 < Mg && trow[0] >= 0 && trow[3] < p.Tout && (idx0 & 3) == 0) {
              typedef unsigned short us4 __attribute__((ext_vector_type(4)));
              *reinterpret_cast<us4*>(y16 + idx0) = us4{f32_to_us<YK>(v[0]), f32_to_us<YK>(v[1]), f32_to_us<YK>(v[2]), f32_to_us<YK>(v[3])};
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (row0 + j < Mg && trow[j] >= 0 && trow[j] < p.Tout)
                  st_act<YK>(p.y, ((size_t)b * M + co[j]) * (size_t)p.Tout + trow[j], v[j]);
            }
          }
        }
      }
      return;
    }
  }
  // ---- epilogue (as conv_gemm_kernel / conv_dma_kernel) ----
  const unsigned rowstride = (unsigned)(p.Tout * P);
  const size_t ybase = ((size_t)b * Mg + m0) * rowstride;
  const float* bias = p.bias ? p.bias + m0 : nullptr;
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
        if (p.res) v += ld_act<YK>(p.res, idx);
        v *= mk;
        if (YB && p.post_scale != 0.f) v *= p.post_scale;
        if (p.accumulate) v += ld_act<YK>(p.y, idx);
        st_act<YK>(p.y, idx, v);
      }
    }
  }
}

// Adds the ks partial slabs of a split launch and applies the epilogue.
__global__ void __launch_bounds__(256) conv_pk_finish_kernel(const VcvConvArgs p, const float* __restrict__ part, int ks) {
  const int U = p.Q * p.P;
  const size_t n = (size_t)p.B * p.Mg * U;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int u = (int)(i % U);
  const size_t bm = i / U;
  const int m = (int)(bm % p.Mg), b = (int)(bm / p.Mg);
  const int q = u / p.P, pc = u - q * p.P;
  const int trow = q * p.os + p.oo;
  if (trow < 0 || trow >= p.Tout) return;
  float v = 0.f;
  for (int k = 0; k < ks; ++k) v += part[(size_t)k * n + i];
  v *= p.alpha;
  if (p.bias) v += p.bias[m];
  v = vcv_act(v, p.out_act, p.slope);
  const size_t idx = (bm * p.Tout + trow) * p.P + pc;
  if (p.out_tf == VCV_TF_DLEAKY) v *= vcv_dleaky(p.oaux[idx], p.slope);
  else if (p.out_tf == VCV_TF_DRELU) v = p.oaux[idx] > 0.f ? v : 0.f;
  else if (p.out_tf == VCV_TF_DTANH) v *= 1.f - p.oaux[idx] * p.oaux[idx];
  if (p.res) v += p.res[idx];
  if (p.mask) v *= p.mask[(size_t)b * p.Tout + trow];
  if (p.accumulate) v += p.y[idx];
  p.y[idx] = v;
}

// The same pass for launches whose output index equals the slab index (os == 1, oo == 0, no mask, element count a
// multiple of four): four consecutive elements per thread, 16-byte loads of every slab and epilogue operand.
__global__ void __launch_bounds__(256) conv_pk_finish4_kernel(const VcvConvArgs p, const float* __restrict__ part, int ks) {
  const int U = p.Q * p.P;
  const size_t n = (size_t)p.B * p.Mg * U;
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  f32x4 v = *reinterpret_cast<const f32x4*>(part + i);
  for (int k = 1; k < ks; ++k) v += *reinterpret_cast<const f32x4*>(part + (size_t)k * n + i);
  f32x4 oa = {0.f, 0.f, 0.f, 0.f}, rr = oa, yy = oa;
  if (p.out_tf >= VCV_TF_DLEAKY) oa = *reinterpret_cast<const f32x4*>(p.oaux + i);
  if (p.res) rr = *reinterpret_cast<const f32x4*>(p.res + i);
  if (p.accumulate) yy = *reinterpret_cast<const f32x4*>(p.y + i);
  const size_t bm0 = i / U;
  const int u0 = (int)(i - bm0 * U);
  f32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const size_t bm = u0 + j < U ? bm0 : bm0 + 1;  // (a group of four may run into the next row)
    float x = p.alpha * v[j];
    if (p.bias) x += p.bias[(int)(bm % p.Mg)];
    x = vcv_act(x, p.out_act, p.slope);
    if (p.out_tf == VCV_TF_DLEAKY) x *= vcv_dleaky(oa[j], p.slope);
    else if (p.out_tf == VCV_TF_DRELU) x = oa[j] > 0.f ? x : 0.f;
    else if (p.out_tf == VCV_TF_DTANH) x *= 1.f - oa[j] * oa[j];
    o[j] = x + rr[j] + yy[j];
  }
  *reinterpret_cast<f32x4*>(p.y + i) = o;
}

}  // namespace

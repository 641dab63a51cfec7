// attention.hip -- fused relative-position self-attention of the content encoder
// (vits/model/transformer/relative_attention_transformer.py:150-182: scores = (q / sqrt(dk)) k^T + banded relative-key
// logits, masked_fill(mask == 0, -1e4), softmax, dropout, p v + banded relative values), forward and backward, one
// launch forward and two backward (row pass, column pass) per layer instead of four and six.
//
// Both contractions run on the matrix cores straight from the [B, C, T] activations -- fp32 inputs:
// v_mfma_f32_32x32x2_f32 (exact fp32); bf16 mode: operands rounded to bf16 on the way in, v_mfma_f32_32x32x16_bf16,
// fp32 accumulate -- with no packing and no materialised transposes:
//   * QK^T (and dP = dO^T V in the backward pass): reduction over the head's channels d; both operands are read as
//     x[d][t .. t + 31], one coalesced 128-byte run per half-wave and reduction step.
//   * P V^T, dQ = dS K^T (reduction over keys j): the probability tile of the workgroup's 32 query rows lives in LDS
//     ([32][TP], TP = 2 mod 64: the fragment reads of the 64 lanes hit 64 different banks), the other operand is read
//     per channel row.
//   * dV = Pd^T dO, dK = dS^T Q (reduction over queries i): column pass, P and dS re-read from HBM (T^2 floats per head).
// Softmax, the banded relative terms, the -1e4 fill and dropout (counter-based mask regenerated from (seed, index): the
// dropped probabilities are never stored for the backward pass) are VALU phases on the LDS tile between the two
// contractions.  The probabilities go to HBM only when the caller wants them (training: the backward pass; `attn`).
#include <type_traits>

#include "common.h"
#include "prof.h"

#ifdef VCV_ATTN_STAMPS
static void* g_attn_stamps = nullptr;
extern "C" void vcv_attn_set_stamps(void* p) { g_attn_stamps = p; }
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wsum_all(float s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}
__device__ __forceinline__ float wmax_all(float s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s = fmaxf(s, __shfl_xor(s, o, 64));
  return s;
}
// the dropout mask of vits_blocks.hip (vcv_dropout / the unfused attention path): same stream, same masks
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.f;
}

struct AttnArgs {
  const float *q, *k, *v, *embk, *embv, *mask, *dO;
  const float* Pin;       // saved probabilities (backward)
  float *out, *P, *Pd;    // forward outputs (P / Pd may be null)
  float *dS, *dq, *dk_, *dv, *dembk, *dembv;
  int B, H, dk, T, w, TP;
  float qscale, pdrop;
  unsigned long long seed;
  const unsigned long long* seed_off;  // non-null while a launch sequence is captured: *seed_off is added to seed (graph replays: version.hip)
};

// One MFMA step: F32: 2 reduction elements (lane half h supplies element 2s + h); BF16: 16 (lane half h supplies
// elements 16s + 8h .. + 7).  val(k) returns the operand element of reduction index k for this lane's row / column.
template <bool BF> struct Op;
template <> struct Op<false> {
  typedef float frag;
  static constexpr int KS = 2;
  template <class F> static __device__ __forceinline__ frag make(int s, int h, F&& val) { return val(2 * s + h); }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};
template <> struct Op<true> {
  typedef bf16x8 frag;
  static constexpr int KS = 16;
  template <class F> static __device__ __forceinline__ frag make(int s, int h, F&& val) {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)val(16 * s + 8 * h + e);
    return v;
  }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

// A row-major fp32 matrix [rows][ld] read through a range-checked buffer descriptor: rows past the end fall outside the
// descriptor and read 0, columns past `cols` are sent out of range by a select -- no branch per load (the first version's
// `cond ? ptr[...] : 0.f` compiled to a branch around every one of a fragment's 32 loads: 28 of the forward kernel's 54 us)
struct Mat {
  __amdgpu_buffer_rsrc_t rs;
  int ld, cols;
};
__device__ __forceinline__ Mat mat(const float* p, int rows, int ld, int cols) {
  Mat m;
  m.rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, rows * ld * 4, 0x00020000);
  m.ld = ld, m.cols = cols;
  return m;
}
__device__ __forceinline__ float ldm(const Mat& m, int r, int c) {
  const unsigned off = c < m.cols ? (unsigned)(r * m.ld + c) * 4u : 0x80000000u;
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(m.rs, off, 0, 0));
}

constexpr int NWV = 8;     // waves per workgroup
constexpr int NTH = 64 * NWV;
constexpr int RELP = 16;   // relative positions held per row (2w + 1 <= 16)
constexpr int OP = 33;     // pitch of the partial-output tiles

// row m of accumulator element e of lane half h (32x32 MFMA output layout; the lane's l31 is the column)
__device__ __forceinline__ int acc_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// T1[32][TP] = alpha * A^T B for the 32 rows i0.. of A: T1[m][j] = alpha * sum_d A[d][i0 + m] * Bm[d][j]  (phase 1 of the
// forward and of the backward row pass).  Key tiles are dealt to the NWV waves.
template <bool BF>
__device__ __forceinline__ void rows_times_keys(const float* __restrict__ A, const float* __restrict__ Bm, float* T1, int i0, int T,
                                                int dk, int TP, float alpha, int wave, int lane) {
  typedef Op<BF> O;
  const int l31 = lane & 31, h = lane >> 5;
  constexpr int MAXS = 64 / O::KS;  // dk <= 64
  const int nks = (dk + O::KS - 1) / O::KS;
  typename O::frag fa[MAXS];
  const int i = i0 + l31;
  const Mat mA = mat(A, dk, T, T), mB = mat(Bm, dk, T, T);
#pragma unroll
  for (int s = 0; s < MAXS; ++s)
    if (s < nks) fa[s] = O::make(s, h, [&](int d) { return ldm(mA, d, i); });
  const int nkt = (T + 31) >> 5;
  typename O::frag fb[MAXS], fn[MAXS];
  auto loadB = [&](typename O::frag (&f)[MAXS], int jt) __attribute__((always_inline)) {
    const int j = jt * 32 + l31;
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
      if (s < nks) f[s] = O::make(s, h, [&](int d) { return ldm(mB, d, j); });
  };
  if (wave < nkt) loadB(fb, wave);
  for (int jt = wave; jt < nkt; jt += NWV) {
    if (jt + NWV < nkt) loadB(fn, jt + NWV);  // the next tile's loads fly under this tile's MFMAs
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < MAXS; ++s)
      if (s < nks) acc = O::mma(fa[s], fb[s], acc);
#pragma unroll
    for (int e = 0; e < 16; ++e) T1[acc_row(e, h) * TP + jt * 32 + l31] = acc[e] * alpha;
#pragma unroll
    for (int s = 0; s < MAXS; ++s) fb[s] = fn[s];
  }
}

constexpr int CK = 128;       // reduction elements of the row-strided operand staged per chunk
constexpr int CKP = CK + 2;   // LDS pitch = 2 mod 64: the 64 lanes of a fragment read hit 64 banks
constexpr int RPP = NTH / CK;  // rows of a chunk staged per pass of the workgroup
constexpr int NRS = 64 / RPP;  // staging registers per thread and chunk
constexpr int BSF = (64 * CKP > NWV * 32 * 33) ? 64 * CKP : NWV * 32 * 33;  // floats of the chunk / partial-tile region

// A chunk of the row-strided operand, Bs[d][0 .. CK) = Bm[d][c0 .. c0 + CK) (zero past T / past the dk rows), in two
// halves so that the loads of chunk c + 1 fly under the MFMAs of chunk c: coalesced global reads along the row into
// registers (NR = 64 / RPP loads per thread, all in flight at once), then the LDS writes.
template <int NR>
__device__ __forceinline__ void load_rows(const float* __restrict__ Bm, float (&v)[NR], int c0, int T, int dk, int tid) {
  const int jj = tid & (CK - 1), dbase = tid >> 7;  // NTH threads = RPP rows x 128 columns per pass
  const int j = c0 + jj;
  const Mat mB = mat(Bm, dk, T, T);
#pragma unroll
  for (int u = 0; u < NR; ++u) v[u] = ldm(mB, RPP * u + dbase, j);
}
template <int NR>
__device__ __forceinline__ void store_rows(float* Bs, const float (&v)[NR], int dk, int tid) {
  const int jj = tid & (CK - 1), dbase = tid >> 7;
  const int rows = (dk + 31) & ~31;
#pragma unroll
  for (int u = 0; u < NR; ++u)
    if (RPP * u + dbase < rows) Bs[(RPP * u + dbase) * CKP + jj] = v[u];
}

// out[d][i0 + m] = alpha * (sum_j T1[m][j] * Bm[d][j] + sum_r T1[m][i0 + m + r - w] * emb[r][d])  (phase 3 of the forward:
// P V^T + relative values; of the backward row pass: dS K^T + relative keys).  Bm is row-strided for the MFMA's lanes
// (lane = channel d), so it goes through LDS in chunks of CK keys (`bs`: 64 * CKP floats; `red` may alias it).  The
// dk / 32 column tiles and the steps of a chunk are dealt to the NWV waves; partial tiles meet in `red`.
template <bool BF>
__device__ __forceinline__ void tile_times_rows(const float* T1, const float* __restrict__ Bm, const float* __restrict__ emb,
                                                float* __restrict__ out, float* bs, float* red, int i0, int T, int dk, int TP, int w,
                                                float alpha, int wave, int lane, int tid) {
  typedef Op<BF> O;
  const int l31 = lane & 31, h = lane >> 5;
  const int nt = (dk + 31) >> 5;         // column tiles (1 or 2)
  const int wpt = NWV / nt;              // waves per column tile
  const int tile = wave % nt, kp = wave / nt;
  const int d = tile * 32 + l31;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const float* Trow = T1 + l31 * TP;
  const float* Brow = bs + d * CKP;
  float stg[NRS];
  load_rows(Bm, stg, 0, T, dk, tid);
  for (int c0 = 0; c0 < T; c0 += CK) {
    __syncthreads();  // the previous chunk's fragment reads are done
    store_rows(bs, stg, dk, tid);
    __syncthreads();
    if (c0 + CK < T) load_rows(Bm, stg, c0 + CK, T, dk, tid);  // in flight under this chunk's MFMAs
    const int nst = ((T - c0 < CK ? T - c0 : CK) + O::KS - 1) / O::KS;
    for (int s = kp; s < nst; s += wpt) {
      const typename O::frag fa = O::make(s, h, [&](int j) { return Trow[c0 + j]; });  // (columns >= T of the tile are zero)
      const typename O::frag fb = O::make(s, h, [&](int j) { return Brow[j]; });
      acc = O::mma(fa, fb, acc);
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) red[(wave * 32 + acc_row(e, h)) * OP + l31] = acc[e];
  __syncthreads();
  const int nr = 2 * w + 1;
  for (int idx = tid; idx < 32 * dk; idx += NTH) {
    const int m = idx & 31, dd = idx >> 5;
    const int i = i0 + m;
    if (i >= T) continue;
    const int tl = dd >> 5, n = dd & 31;
    float s = 0.f;
    for (int k2 = 0; k2 < wpt; ++k2) s += red[((k2 * nt + tl) * 32 + m) * OP + n];
    for (int r = 0; r < nr; ++r) {
      const int j = i + r - w;
      if (j >= 0 && j < T) s += T1[m * TP + j] * emb[r * dk + dd];
    }
    out[(size_t)dd * T + i] = s * alpha;
  }
}

constexpr int QP = 33;  // pitch of the staged 32-position operand tile

// rel[m][r] = alpha * sum_d A[d][i0 + m] * emb[r][d], through LDS: `at` [dk][QP] takes the 32-position tile of A (coalesced
// loads, all in flight at once), `es` [nr][dk] the table; then one thread per (m, r)
__device__ __forceinline__ void band_dots(const float* __restrict__ A, const float* __restrict__ emb, float* rel, float* at, float* es,
                                          int i0, int T, int dk, int nr, float alpha, int tid) {
  constexpr int RP = NTH / 32, NU = 64 / RP;  // channel rows per pass, passes
  const int m = tid & 31, dq = tid >> 5;
  float v[NU];
  const Mat mA = mat(A, dk, T, T);
#pragma unroll
  for (int u = 0; u < NU; ++u) v[u] = ldm(mA, dq + RP * u, i0 + m);
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (dq + RP * u < dk) at[(dq + RP * u) * QP + m] = v[u];
  for (int idx = tid; idx < nr * dk; idx += NTH) es[idx] = emb[idx];
  __syncthreads();
  for (int r = dq; r < nr; r += RP) {
    float s = 0.f;
    for (int d = 0; d < dk; ++d) s += at[d * QP + m] * es[r * dk + d];
    rel[m * RELP + r] = s * alpha;
  }
}

template <bool BF>
__global__ void __launch_bounds__(NTH) rel_attn_fwd_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T, dk = p.dk, TP = p.TP, nr = 2 * p.w + 1;
  float* S = sm;
  float* rel = S + 32 * TP;
  float* msk = rel + 32 * RELP;   // the batch element's mask row
  float* etab = msk + TP;         // the table of the second contraction's band term
  float* bs = etab + RELP * 64;   // staged chunk of the row-strided operand; before that the tile / table of the band
                                  // logits, after it the partial-output tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.y, b = g / p.H, i0 = blockIdx.x * 32;
  const float* qg = p.q + (size_t)g * dk * T;
  const float* kg = p.k + (size_t)g * dk * T;
  const float* vg = p.v + (size_t)g * dk * T;
  const float* mrow = msk;
  for (int j = tid; j < T; j += NTH) msk[j] = p.mask[(size_t)b * T + j];
  for (int idx = tid; idx < nr * dk; idx += NTH) etab[idx] = p.embv[idx];

  band_dots(qg, p.embk, rel, bs, bs + 64 * QP, i0, T, dk, nr, p.qscale, tid);
  rows_times_keys<BF>(qg, kg, S, i0, T, dk, TP, p.qscale, wave, lane);
  __syncthreads();
  // softmax of the wave's 32 / NWV rows
  const float inv_keep = p.pdrop > 0.f ? 1.f / (1.f - p.pdrop) : 1.f;
  for (int rr = 0; rr < 32 / NWV; ++rr) {
    const int m = wave * (32 / NWV) + rr, i = i0 + m;
    float* Sr = S + m * TP;
    if (i >= T) {
      for (int j = lane; j < TP; j += 64) Sr[j] = 0.f;
      continue;
    }
    const float mi = mrow[i];
    float mx = -INFINITY;
    for (int j = lane; j < T; j += 64) {
      float v = Sr[j];
      const int r = j - i + p.w;
      if (r >= 0 && r < nr) v += rel[m * RELP + r];
      if (mi * mrow[j] == 0.f) v = -1e4f;
      Sr[j] = v;
      mx = fmaxf(mx, v);
    }
    mx = wmax_all(mx);
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) {
      const float e = __expf(Sr[j] - mx);
      Sr[j] = e;
      sum += e;
    }
    sum = wsum_all(sum);
    const float inv = 1.f / sum;
    const size_t rowoff = ((size_t)g * T + i) * T;
    for (int j = lane; j < TP; j += 64) {
      float pd = 0.f;
      if (j < T) {
        const float pv = Sr[j] * inv;
        if (p.P) p.P[rowoff + j] = pv;
        pd = p.pdrop > 0.f ? pv * drop_scale(p.seed + (p.seed_off ? *p.seed_off : 0ull), rowoff + j, p.pdrop, inv_keep) : pv;
        if (p.Pd) p.Pd[rowoff + j] = pd;
      }
      Sr[j] = pd;
    }
  }
  __syncthreads();
  tile_times_rows<BF>(S, vg, etab, p.out + (size_t)g * dk * T, bs, bs, i0, T, dk, TP, p.w, 1.f, wave, lane, tid);
}

// Backward, row pass: per (head, 32 query rows): dPd = dO^T V + band, dS = Pd * dPd - P * sum_j(Pd * dPd) (zero where
// masked), dS -> HBM, dQ = qscale * (dS K^T + band), and this tile's share of the two relative-position table gradients.
template <bool BF>
__global__ void __launch_bounds__(NTH) rel_attn_bwd_rows_kernel(const AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T, dk = p.dk, TP = p.TP, nr = 2 * p.w + 1;
  float* D = sm;
  float* rel = D + 32 * TP;
  float* msk = rel + 32 * RELP;
  float* etab = msk + TP;
  float* bs = etab + RELP * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = blockIdx.y, b = g / p.H, i0 = blockIdx.x * 32;
  const float* qg = p.q + (size_t)g * dk * T;
  const float* kg = p.k + (size_t)g * dk * T;
  const float* vg = p.v + (size_t)g * dk * T;
  const float* og = p.dO + (size_t)g * dk * T;
  const float* mrow = msk;
  const float inv_keep = p.pdrop > 0.f ? 1.f / (1.f - p.pdrop) : 1.f;
  for (int j = tid; j < T; j += NTH) msk[j] = p.mask[(size_t)b * T + j];
  for (int idx = tid; idx < nr * dk; idx += NTH) etab[idx] = p.embk[idx];

  band_dots(og, p.embv, rel, bs, bs + 64 * QP, i0, T, dk, nr, 1.f, tid);
  rows_times_keys<BF>(og, vg, D, i0, T, dk, TP, 1.f, wave, lane);
  __syncthreads();
  for (int rr = 0; rr < 32 / NWV; ++rr) {
    const int m = wave * (32 / NWV) + rr, i = i0 + m;
    float* Dr = D + m * TP;
    if (i >= T) {
      for (int j = lane; j < TP; j += 64) Dr[j] = 0.f;
      continue;
    }
    const size_t rowoff = ((size_t)g * T + i) * T;
    float dot = 0.f;
    for (int j = lane; j < T; j += 64) {
      float v = Dr[j];
      const int r = j - i + p.w;
      if (r >= 0 && r < nr) v += rel[m * RELP + r];
      Dr[j] = v;
      const float pv = p.Pin[rowoff + j];
      const float pd = p.pdrop > 0.f ? pv * drop_scale(p.seed + (p.seed_off ? *p.seed_off : 0ull), rowoff + j, p.pdrop, inv_keep) : pv;
      dot += v * pd;
    }
    dot = wsum_all(dot);
    const float mi = mrow[i];
    for (int j = lane; j < TP; j += 64) {
      float ds = 0.f;
      if (j < T) {
        const float pv = p.Pin[rowoff + j];
        const float pd = p.pdrop > 0.f ? pv * drop_scale(p.seed + (p.seed_off ? *p.seed_off : 0ull), rowoff + j, p.pdrop, inv_keep) : pv;
        ds = pd * Dr[j] - pv * dot;
        if (mi * mrow[j] == 0.f) ds = 0.f;
        p.dS[rowoff + j] = ds;
      }
      Dr[j] = ds;
    }
  }
  __syncthreads();
  // table gradients of this row tile: dembk[r][d] += qscale * sum_i dS[i][i+r-w] q[d][i];  dembv[r][d] += sum_i Pd[i][i+r-w] dO[d][i]
  for (int idx = tid; idx < nr * dk; idx += NTH) {
    const int r = idx / dk, d = idx - r * dk;
    float ek = 0.f, ev = 0.f;
    for (int m = 0; m < 32; ++m) {
      const int i = i0 + m, j = i + r - p.w;
      if (i >= T || j < 0 || j >= T) continue;
      const size_t off = ((size_t)g * T + i) * T + j;
      const float pv = p.Pin[off];
      const float pd = p.pdrop > 0.f ? pv * drop_scale(p.seed + (p.seed_off ? *p.seed_off : 0ull), off, p.pdrop, inv_keep) : pv;
      ek += D[m * TP + j] * qg[(size_t)d * T + i];
      ev += pd * og[(size_t)d * T + i];
    }
    unsafeAtomicAdd(p.dembk + idx, ek * p.qscale);
    unsafeAtomicAdd(p.dembv + idx, ev);
  }
  tile_times_rows<BF>(D, kg, etab, p.dq + (size_t)g * dk * T, bs, bs, i0, T, dk, TP, p.w, p.qscale, wave, lane, tid);
}

// Backward, column pass: per (head, 32 keys): dV[d][j] = sum_i Pd[i][j] dO[d][i], dK[d][j] = qscale * sum_i dS[i][j] q[d][i].
template <bool BF>
__global__ void __launch_bounds__(NTH) rel_attn_bwd_cols_kernel(const AttnArgs p) {
  typedef Op<BF> O;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* os = sm;               // dO chunk [64][CKP]
  float* qs = sm + BSF;         // q chunk
  float* red0 = sm;             // the partial tiles reuse the chunk buffers (NWV * 32 * OP floats each)
  float* red1 = sm + BSF;
  const int T = p.T, dk = p.dk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int g = blockIdx.y, j0 = blockIdx.x * 32;
  const float* qg = p.q + (size_t)g * dk * T;
  const float* og = p.dO + (size_t)g * dk * T;
  const float inv_keep = p.pdrop > 0.f ? 1.f / (1.f - p.pdrop) : 1.f;
  const int nt = (dk + 31) >> 5;
  const int wpt = NWV / nt;
  const int tile = wave % nt, kp = wave / nt;
  const int d = tile * 32 + l31, j = j0 + l31;
  const float* orow = os + d * CKP;
  const float* qrow = qs + d * CKP;
  f32x16 av, ak;
#pragma unroll
  for (int e = 0; e < 16; ++e) av[e] = ak[e] = 0.f;
  const Mat mP = mat(p.Pin + (size_t)g * T * T, T, T, T), mS = mat(p.dS + (size_t)g * T * T, T, T, T);
  float so[NRS], sq[NRS];
  load_rows(og, so, 0, T, dk, tid);
  load_rows(qg, sq, 0, T, dk, tid);
  for (int c0 = 0; c0 < T; c0 += CK) {
    __syncthreads();
    store_rows(os, so, dk, tid);
    store_rows(qs, sq, dk, tid);
    __syncthreads();
    if (c0 + CK < T) {
      load_rows(og, so, c0 + CK, T, dk, tid);
      load_rows(qg, sq, c0 + CK, T, dk, tid);
    }
    const int nst = ((T - c0 < CK ? T - c0 : CK) + O::KS - 1) / O::KS;
    for (int s = kp; s < nst; s += wpt) {
      const typename O::frag fp = O::make(s, h, [&](int ii) {
        const int i = c0 + ii;
        const float pv = ldm(mP, i, j);  // (0 past the matrix)
        return p.pdrop > 0.f ? pv * drop_scale(p.seed + (p.seed_off ? *p.seed_off : 0ull), ((size_t)g * T + i) * T + j, p.pdrop, inv_keep) : pv;
      });
      const typename O::frag fs = O::make(s, h, [&](int ii) { return ldm(mS, c0 + ii, j); });
      const typename O::frag fo = O::make(s, h, [&](int ii) { return orow[ii]; });
      const typename O::frag fq = O::make(s, h, [&](int ii) { return qrow[ii]; });
      av = O::mma(fp, fo, av);
      ak = O::mma(fs, fq, ak);
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    red0[(wave * 32 + acc_row(e, h)) * OP + l31] = av[e];
    red1[(wave * 32 + acc_row(e, h)) * OP + l31] = ak[e];
  }
  __syncthreads();
  for (int idx = tid; idx < 32 * dk; idx += NTH) {
    const int m = idx & 31, dd = idx >> 5;
    if (j0 + m >= T) continue;
    const int tl = dd >> 5, n = dd & 31;
    float sv = 0.f, sk = 0.f;
    for (int k2 = 0; k2 < wpt; ++k2) {
      const int wv = k2 * nt + tl;
      sv += red0[(wv * 32 + m) * OP + n];
      sk += red1[(wv * 32 + m) * OP + n];
    }
    p.dv[((size_t)g * dk + dd) * T + j0 + m] = sv;
    p.dk_[((size_t)g * dk + dd) * T + j0 + m] = sk * p.qscale;
  }
}

// =====================================================================================================================
// Forward, round 4: ONE WAVE PER 32 QUERY ROWS, the whole score tile of those rows in registers (T <= 256, 32 or 64
// channels per head).
//
// The kernel above gives a 32-row tile to a workgroup of 8 waves and walks through five barrier-separated phases with the
// tile in LDS; at the content encoder's sizes (T ~ 200, 64 channels per head) each phase is a few microseconds of
// latency and the matrix pipe ran 5-15 % of the time.  Here a wave owns its 32 queries from the first load to the output
// store and never meets another wave after the prologue:
//   * scores TRANSPOSED: S^T tile = K^T-tile (A: rows = keys) x Q (B: columns = queries), so in the accumulator layout a
//     lane IS a query (column l31) and its registers are that query's keys (rows (e & 3) + 8 (e >> 2) + 4 h of tile jt):
//     the softmax is a reduction over the lane's own registers plus ONE exchange with lane ^ 32 -- no LDS, no barrier;
//   * P V: O^T[d][i] = sum_j V[d][j] Pd[i][j] with B = the probability registers AS THEY ARE (reduction index of step s',
//     half h := key (s' & 3) + 8 (s' >> 2) + 4 h of the tile -- any order of the reduction is fine as long as A uses the
//     same one) and A = V read from an LDS image [d][j] of odd pitch (one conflict-free ds_read_b32 per MFMA); V is the
//     only operand staged through LDS, once per workgroup of FW waves (the prologue's single barrier);
//   * the two banded relative-position products ride on the matrix cores too, always in exact fp32
//     (v_mfma_f32_32x32x2_f32): R^T = embk (9 rows of a 32-row A tile) x Q before the softmax, embv^T x band(Pd) after
//     it; the band of a lane is exchanged through a wave-private [32][17] LDS scratch;
//   * probabilities (training: the backward pass reads P; `attn`) leave through a wave-private 32 x 33 transpose tile so
//     that the global stores are 128-byte runs of one row.
// K goes through LDS the same way ([d][j]: the A fragments of S^T read consecutive keys per lane -- conflict-free): fragments
// straight from global memory, double-buffered one key tile ahead, left each tile's MFMAs waiting for an L2 round trip.
constexpr int FW = 4;    // waves = query tiles per workgroup
constexpr int RWP = 17;  // pitch of the wave-private [32 queries][relative position] scratch
constexpr int PTP = 33;  // pitch of the wave-private 32 x 32 transpose tile

// compile-time loop (indices stay literals whatever the unroll thresholds say: the accumulator tiles must be registers)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ f32x16 mma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// rows of a [DK][T] fp32 matrix at one column per lane: byte offset = col_off (0x80000000 for a column >= T: out of range,
// reads 0) + row * T4, the row term a scalar -- one v_add per load, no multiply, no compare
struct ColLoader {
  __amdgpu_buffer_rsrc_t rs;
  unsigned T4;
  __device__ __forceinline__ float operator()(unsigned col_off, int row) const {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, col_off + (unsigned)row * T4, 0, 0));
  }
};
__device__ __forceinline__ ColLoader col_loader(const float* p, int rows, int T) {
  ColLoader c;
  c.rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, rows * T * 4, 0x00020000);
  c.T4 = (unsigned)T * 4u;
  return c;
}

// diagnostic build only (-DVCV_ATTN_STAMPS, tools/probes/attn_stamps.py): 100 MHz time stamps at the phase boundaries of
// every wave into a buffer of their own; the product build executes none of this
#ifdef VCV_ATTN_STAMPS
#define VCV_STAMP(k)                                                                                              \
  do {                                                                                                            \
    if (p.Pd && lane == 0)                                                                                        \
      ((unsigned long long*)p.Pd)[(((size_t)blockIdx.y * gridDim.x + blockIdx.x) * FW + wave) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define VCV_STAMP(k)
#endif

template <bool BF, int NKT, int DK>
__global__ void __launch_bounds__(64 * FW) rel_attn_fwd_rows_kernel(const AttnArgs p) {
  typedef Op<BF> O;
  constexpr int NS = DK / O::KS;  // reduction steps of the channel contraction
  constexpr int NT = DK / 32;     // 32-channel output tiles
  constexpr int TPV = NKT * 32 + 1;
  constexpr float LOG2E = 1.4426950408889634f;
  static_assert(NKT * 32 == 64 * FW, "one staged column per thread");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T, w = p.w, nr = 2 * p.w + 1;
  const int nkt = (T + 31) >> 5;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* bias = sm;                // [NKT * 32]: 0 for a key < T, -inf past the end (16-byte aligned: read four at a time)
  float* Vs = bias + NKT * 32;     // [DK][TPV] (columns >= T: zero)
  float* Ks = Vs + DK * TPV;       // [DK][TPV]
  float* etab = Ks + DK * TPV;     // embv as [RELP][DK], rows >= nr zero
  float* relw = etab + RELP * DK + wave * (32 * RWP + 32 * PTP);  // wave-private: [32][RWP] ...
  float* ptile = relw + 32 * RWP;                                  // ... and [32][PTP]
  const int g = blockIdx.y, b = g / p.H;
  const int it = blockIdx.x * FW + wave;  // this wave's query tile (>= nkt: an idle wave that only helps staging)
  const ColLoader lq = col_loader(p.q + (size_t)g * DK * T, DK, T), lk = col_loader(p.k + (size_t)g * DK * T, DK, T),
                  lv = col_loader(p.v + (size_t)g * DK * T, DK, T);

  VCV_STAMP(0);
  // ---- prologue: this lane's Q column (fp32, lane half h holds channels 2 s + h) and its row of the relative-key table, then
  // the K column of this THREAD (staging: thread = key column, all DK rows in flight); the mask words and the R^T MFMAs
  // run under the K loads, the V loads under the S^T MFMAs
  const int i = it * 32 + l31;  // this lane's query (>= T in the last tile or an idle wave: loads come back 0, nothing is stored)
  const unsigned ioff = i < T ? (unsigned)i * 4u : 0x80000000u;
  float q32[DK / 2];
#pragma unroll
  for (int s = 0; s < DK / 2; ++s) q32[s] = lq(ioff + h * lq.T4, 2 * s);
  float ea[DK / 2];  // row r = l31 of the relative-key table (A operand of R^T)
  {
    const float* erow = p.embk + (l31 < nr ? l31 : 0) * DK + h;
#pragma unroll
    for (int s = 0; s < DK / 2; ++s) ea[s] = erow[2 * s];
  }
  const unsigned joff = tid < T ? (unsigned)tid * 4u : 0x80000000u;
  float stg[DK];
#pragma unroll
  for (int u = 0; u < DK; ++u) stg[u] = lk(joff, u);

  // key-validity mask of the batch element as wave-wide bit masks (64 keys each); keys >= T count as unmasked here.  A
  // masked QUERY masks its whole row: folded into the lane's copy of the bits
  const float mi = i < T ? p.mask[(size_t)b * T + i] : 1.f;
  unsigned long long mb[NKT / 2];
  bool allone = true;
#pragma unroll
  for (int c = 0; c < NKT / 2; ++c) {
    const int j = 64 * c + lane;
    const float m = j < T ? p.mask[(size_t)b * T + j] : 1.f;
    mb[c] = __ballot(m != 0.f);
    allone = allone && mb[c] == ~0ull;
  }
  bias[tid] = tid < T ? 0.f : -INFINITY;
  for (int idx = tid; idx < RELP * DK; idx += 64 * FW) etab[idx] = idx < nr * DK ? p.embv[idx] : 0.f;

  // ---- relative-key logits R^T = embk x Q (exact fp32): row r of the tile = relative position r
  {
    f32x16 racc;
#pragma unroll
    for (int e = 0; e < 16; ++e) racc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < DK / 2; ++s) racc = mma32(l31 < nr ? ea[s] : 0.f, q32[s], racc);
    // wave-private scratch row of this query: slot 1 + r = logit of relative position r; slots 0 and RWP - 1 stay zero, so a
    // gather with the index clamped into [0, RWP - 1] needs no select (and a clamped scatter later dumps into them)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = acc_row(e, h);  // rows >= nr of the tile are zero (A rows l31 >= nr were zeroed)
      if (r < RWP - 2) relw[l31 * RWP + 1 + r] = racc[e];
    }
    if (h == 0) relw[l31 * RWP] = 0.f; else relw[l31 * RWP + RWP - 1] = 0.f;
  }
#pragma unroll
  for (int u = 0; u < DK; ++u) Ks[u * TPV + tid] = stg[u];
#pragma unroll
  for (int u = 0; u < DK; ++u) stg[u] = lv(joff, u);
  __syncthreads();
  VCV_STAMP(1);
  VCV_STAMP(2);

  // ---- S^T = K^T Q: accumulators acc[jt][e] = score of (query i, key jt * 32 + acc_row(e, h))
  typename O::frag qo[NS];
  if constexpr (BF) {
#pragma unroll
    for (int s = 0; s < NS; ++s) qo[s] = O::make(s, h, [&](int d) { return lq(ioff, d); });
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) qo[s] = q32[s];
  }
  f32x16 acc[NKT];
  static_for<0, NKT>([&](auto JT) {
    constexpr int jt = decltype(JT)::value;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[jt][e] = 0.f;
  });
  const float* Kl = Ks + (BF ? 8 * h : h) * TPV + l31;  // A fragments: rows = keys (lane l31), channels by lane half
  static_for<0, NKT>([&](auto JT) {
    constexpr int jt = decltype(JT)::value;
    if (jt < nkt && it < nkt) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        typename O::frag a;
        if constexpr (BF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = (__bf16)Kl[(16 * s + e) * TPV + jt * 32];
        } else {
          a = Kl[2 * s * TPV + jt * 32];
        }
        acc[jt] = O::mma(a, qo[s], acc[jt]);
      }
    }
  });
#pragma unroll
  for (int u = 0; u < DK; ++u) Vs[u * TPV + tid] = stg[u];
  __syncthreads();
  if (it >= nkt) return;
  VCV_STAMP(3);

  // ---- everything after the scores, twice: MASKED = false when every key and query of the batch element is valid (the
  // content encoder's training batches: the mask is all ones there) skips the -1e4 fill.  ONE uniform branch picks a copy and
  // the two never join again: a uniform `if` around work on the accumulator tiles inside a straight-line phase made the
  // compiler shuttle all of them between AGPRs and VGPRs at every join (2,560 accvgpr reads, 9 us in the softmax phase).
  auto rest = [&](auto MASKED) __attribute__((always_inline)) {
    constexpr bool masked = decltype(MASKED)::value;
    // scores in the log2 domain: v = (S + R) * qscale * log2(e) (+ -inf for a key past T, from the bias table); row maximum
    const float qs = p.qscale * LOG2E;
    float mx = -INFINITY;
    static_for<0, NKT>([&](auto JT) {
      constexpr int jt = decltype(JT)::value;
      float rv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) rv[e] = 0.f;
      if (jt + 1 >= it && jt <= it + 1) {  // (a branch that leaves the accumulators alone is cheap)
        const int r1 = jt * 32 + 4 * h - i + w + 1;  // 1 + relative position of key row 0
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned idx = (unsigned)(r1 + (e & 3) + 8 * (e >> 2));  // (negative: wraps, clamps to the zero slot at the end)
          rv[e] = relw[l31 * RWP + (idx < (unsigned)(RWP - 1) ? idx : (unsigned)(RWP - 1))];
        }
      }
      unsigned wl = 0;
      if constexpr (masked) {
        wl = (unsigned)(mb[jt >> 1] >> ((jt & 1) * 32)) >> (4 * h);  // bit ce: key row ce + 4 h of this tile unmasked
        wl = mi != 0.f ? wl : 0u;
      }
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const f32x4 bz = *reinterpret_cast<const f32x4*>(bias + jt * 32 + 8 * e4 + 4 * h);
#pragma unroll
        for (int e1 = 0; e1 < 4; ++e1) {
          const int e = 4 * e4 + e1, ce = e1 + 8 * e4;
          float v = (acc[jt][e] + rv[e]) * qs;
          if constexpr (masked) v = (wl & (1u << ce)) ? v : -1e4f * LOG2E;
          v += bz[e1];
          acc[jt][e] = v;
          mx = fmaxf(mx, v);
        }
      }
    });
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
    static_for<0, NKT>([&](auto JT) {
      constexpr int jt = decltype(JT)::value;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float ex = __builtin_amdgcn_exp2f(acc[jt][e] - mx);
        acc[jt][e] = ex;
        sum += ex;
      }
    });
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;  // (stays factored out: everything below is linear in the probabilities; applied at the output)

    VCV_STAMP(4);
    // ---- probability stores through the transpose tile (training: the backward pass reads P; `attn`) and dropout (mask
    // regenerated from (seed, element index)); each behind one uniform branch around its whole tile loop
    auto store_tiles = [&](float* dst) __attribute__((always_inline)) {
      static_for<0, NKT>([&](auto JT) {
        constexpr int jt = decltype(JT)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) ptile[l31 * PTP + (e & 3) + 8 * (e >> 2) + 4 * h] = acc[jt][e] * inv;
        const int jj = jt * 32 + l31;
        float* drow = dst + ((size_t)g * T + it * 32 + h) * T + jj;
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) {
          const float v = ptile[(2 * k2 + h) * PTP + l31];
          if (it * 32 + 2 * k2 + h < T && jj < T) drow[(size_t)(2 * k2) * T] = v;
        }
      });
    };
    if (p.P) store_tiles(p.P);
    if (p.pdrop > 0.f) {
      const float inv_keep = 1.f / (1.f - p.pdrop);
      const unsigned long long seed = p.seed + (p.seed_off ? *p.seed_off : 0ull);
      const size_t rowoff = ((size_t)g * T + i) * T + 4 * h;
      static_for<0, NKT>([&](auto JT) {
        constexpr int jt = decltype(JT)::value;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[jt][e] *= drop_scale(seed, rowoff + jt * 32 + (e & 3) + 8 * (e >> 2), p.pdrop, inv_keep);
      });
    }
    if (p.Pd) store_tiles(p.Pd);

    // ---- band of the (dropped, un-normalised) probabilities: band[r] = Pd[i][i + r - w] into slot 1 + r of the scratch row
    // (every element of the three tiles around the diagonal stores, with the index clamped: off-band elements land in the
    // two dump slots)
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) relw[l31 * RWP + 1 + 8 * h + r8] = 0.f;
    static_for<0, NKT>([&](auto JT) {
      constexpr int jt = decltype(JT)::value;
      if (jt + 1 >= it && jt <= it + 1) {
        const int r1 = jt * 32 + 4 * h - i + w + 1;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned idx = (unsigned)(r1 + (e & 3) + 8 * (e >> 2));
          relw[l31 * RWP + (idx < (unsigned)(RWP - 1) ? idx : (unsigned)(RWP - 1))] = acc[jt][e];
        }
      }
    });

    VCV_STAMP(5);
    // ---- O^T = V Pd^T (+ embv^T band)
    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;
    const float* Vl = Vs + l31 * TPV + 4 * h;
    static_for<0, NKT>([&](auto JT) {
      constexpr int jt = decltype(JT)::value;
      if (jt < nkt) {
        if constexpr (BF) {
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2) {
            bf16x8 bfr;
#pragma unroll
            for (int e = 0; e < 8; ++e) bfr[e] = (__bf16)acc[jt][8 * m2 + e];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              bf16x8 afr;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int s2 = 8 * m2 + e;
                afr[e] = (__bf16)Vl[32 * t * TPV + jt * 32 + (s2 & 3) + 8 * (s2 >> 2)];
              }
              oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, oacc[t], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
              oacc[t] = mma32(Vl[32 * t * TPV + jt * 32 + (s2 & 3) + 8 * (s2 >> 2)], acc[jt][s2], oacc[t]);
          }
        }
      }
    });
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (2 * s < nr) {
        const float bfr = relw[l31 * RWP + 1 + 2 * s + h];  // band[r = 2 s + h] of this lane's query (r >= nr: times a zero table row)
#pragma unroll
        for (int t = 0; t < NT; ++t) oacc[t] = mma32(etab[(2 * s + h) * DK + 32 * t + l31], bfr, oacc[t]);
      }
    VCV_STAMP(6);
    if (i < T) {
      float* og = p.out + (size_t)g * DK * T + i;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) og[(size_t)(32 * t + acc_row(e, h)) * T] = oacc[t][e] * inv;
    }
    VCV_STAMP(7);
  };
  if (allone && __ballot(mi == 0.f) == 0ull) rest(std::false_type());
  else rest(std::true_type());
}

// =====================================================================================================================
// Backward, round 4, in the forward's style (T <= 256, 32 / 64 channels per head): a ROW pass with lane = query and a
// COLUMN pass with lane = key, every contraction with the probability-shaped operand taken from registers.
//
// Row pass (one wave per 32 queries): dPd^T = V^T dO on the matrix cores (A = V image in LDS, B = this lane's dO column) plus the
// banded embv x dO product (the forward's R^T with other operands); P of the lane's rows arrives through the transpose tile
// (coalesced 128-byte row loads); dot_i = sum_j dPd Pd and dS = P (c dPd - dot_i) (c: dropout factor) stay in registers;
// dS leaves through the transpose tile for the column pass; dQ^T = K dS^T + embk^T band(dS) is the forward's P V with K in
// the LDS image (the image buffer is re-staged: V first, K after the dPd MFMAs); the two table gradients are small MFMA
// products band^T x Q^T / band^T x dO^T over the wave's 32 queries, added to HBM with atomics (as before).
// Column pass (one wave per 32 keys): P and dS tiles are loaded with lane = key (rows of the matrices: coalesced), dropout
// regenerated, dV^T = dO Pd and dK^T = Q dS accumulated over the query tiles with A = the LDS images of dO / Q.
template <bool BF, int NKT, int DK>
__global__ void __launch_bounds__(64 * FW) rel_attn_bwd_rows2_kernel(const AttnArgs p) {
  typedef Op<BF> O;
  constexpr int NS = DK / O::KS, NT = DK / 32, TPV = NKT * 32 + 1;
  static_assert(NKT * 32 == 64 * FW, "one staged column per thread");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T, w = p.w, nr = 2 * p.w + 1;
  const int nkt = (T + 31) >> 5;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* Xs = sm;                  // [DK][TPV]: V, later K
  float* etab = Xs + DK * TPV;     // embk as [RELP][DK], rows >= nr zero
  float* relw = etab + RELP * DK + wave * (3 * 32 * RWP + 32 * PTP);  // wave-private band scratch rows: embv x dO ...
  float* relp = relw + 32 * RWP;                                      // ... band(Pd) ...
  float* rels = relp + 32 * RWP;                                      // ... band(dS) ...
  float* ptile = rels + 32 * RWP;                                     // ... and the 32 x 32 transpose tile
  const int g = blockIdx.y, b = g / p.H;
  const int it = blockIdx.x * FW + wave;
  const ColLoader lq = col_loader(p.q + (size_t)g * DK * T, DK, T), lk = col_loader(p.k + (size_t)g * DK * T, DK, T),
                  lv = col_loader(p.v + (size_t)g * DK * T, DK, T), lo = col_loader(p.dO + (size_t)g * DK * T, DK, T);
  const ColLoader lp = col_loader(p.Pin + (size_t)g * T * T, T, T);  // P of this head: rows = queries, columns = keys

  VCV_STAMP(0);
  const int i = it * 32 + l31;
  const unsigned ioff = i < T ? (unsigned)i * 4u : 0x80000000u;
  float do32[DK / 2];
#pragma unroll
  for (int s = 0; s < DK / 2; ++s) do32[s] = lo(ioff + h * lo.T4, 2 * s);
  // dot_i = sum_j dPd[i][j] Pd[i][j] = sum_d dO[d][i] out[d][i] (the forward's output, band values included: both sides are
  // sum_d dO[d][i] (sum_j Pd[i][j] (V[d][j] + band table))): known BEFORE the first tile, so dS is formed tile by tile in one pass
  float dot = 0.f;
  {
    const ColLoader lout = col_loader(p.out + (size_t)g * DK * T, DK, T);
#pragma unroll
    for (int s = 0; s < DK / 2; ++s) dot += do32[s] * lout(ioff + h * lout.T4, 2 * s);
    dot += __shfl_xor(dot, 32, 64);
  }
  float ea[DK / 2];
  {
    const float* erow = p.embv + (l31 < nr ? l31 : 0) * DK + h;
#pragma unroll
    for (int s = 0; s < DK / 2; ++s) ea[s] = erow[2 * s];
  }
  const unsigned joff = tid < T ? (unsigned)tid * 4u : 0x80000000u;
  float stg[DK];
#pragma unroll
  for (int u = 0; u < DK; ++u) stg[u] = lv(joff, u);
  const float mi = i < T ? p.mask[(size_t)b * T + i] : 1.f;
  unsigned long long mb[NKT / 2];
  bool allone = true;
#pragma unroll
  for (int c = 0; c < NKT / 2; ++c) {
    const int j = 64 * c + lane;
    const float m = j < T ? p.mask[(size_t)b * T + j] : 1.f;
    mb[c] = __ballot(m != 0.f);
    allone = allone && mb[c] == ~0ull;
  }
  for (int idx = tid; idx < RELP * DK; idx += 64 * FW) etab[idx] = idx < nr * DK ? p.embk[idx] : 0.f;
  {  // band part of dPd: slot 1 + r of the scratch row = sum_d embv[r][d] dO[d][i]
    f32x16 racc;
#pragma unroll
    for (int e = 0; e < 16; ++e) racc[e] = 0.f;
#pragma unroll
    for (int s = 0; s < DK / 2; ++s) racc = mma32(l31 < nr ? ea[s] : 0.f, do32[s], racc);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int r = acc_row(e, h);
      if (r < RWP - 2) relw[l31 * RWP + 1 + r] = racc[e];
    }
    if (h == 0) relw[l31 * RWP] = 0.f; else relw[l31 * RWP + RWP - 1] = 0.f;
  }
#pragma unroll
  for (int u = 0; u < DK; ++u) Xs[u * TPV + tid] = stg[u];
#pragma unroll
  for (int u = 0; u < DK; ++u) stg[u] = lk(joff, u);
  __syncthreads();
  VCV_STAMP(1);

  // ---- dPd^T = V^T dO
  typename O::frag dob[NS];
  if constexpr (BF) {
#pragma unroll
    for (int s = 0; s < NS; ++s) dob[s] = O::make(s, h, [&](int d) { return lo(ioff, d); });
  } else {
#pragma unroll
    for (int s = 0; s < NS; ++s) dob[s] = do32[s];
  }
  f32x16 acc[NKT];
  static_for<0, NKT>([&](auto JT) {
    constexpr int jt = decltype(JT)::value;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[jt][e] = 0.f;
  });
  const float* Xl = Xs + (BF ? 8 * h : h) * TPV + l31;
  static_for<0, NKT>([&](auto JT) {
    constexpr int jt = decltype(JT)::value;
    if (jt < nkt && it < nkt) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        typename O::frag a;
        if constexpr (BF) {
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = (__bf16)Xl[(16 * s + e) * TPV + jt * 32];
        } else {
          a = Xl[2 * s * TPV + jt * 32];
        }
        acc[jt] = O::mma(a, dob[s], acc[jt]);
      }
    }
  });
  VCV_STAMP(2);
  __syncthreads();  // every wave is done with the V image
#pragma unroll
  for (int u = 0; u < DK; ++u) Xs[u * TPV + tid] = stg[u];
  __syncthreads();  // the K image is there
  if (it >= nkt) return;
  VCV_STAMP(3);

  auto rest = [&](auto MASKED) __attribute__((always_inline)) {
    constexpr bool masked = decltype(MASKED)::value;
    const float inv_keep = p.pdrop > 0.f ? 1.f / (1.f - p.pdrop) : 1.f;
    const size_t rowoff = ((size_t)g * T + i) * T + 4 * h;
    // ---- one pass over the key tiles: dPd = MFMA + band; P through the transpose tile (loads issued two tiles ahead); c =
    // dropout factor; dS = P (c dPd - dot), zero where masked, kept in the accumulators for dQ and sent out through the
    // transpose tile for the column pass; band(Pd) and band(dS) into their scratch rows
#pragma unroll
    for (int r8 = 0; r8 < 8; ++r8) relp[l31 * RWP + 1 + 8 * h + r8] = 0.f, rels[l31 * RWP + 1 + 8 * h + r8] = 0.f;
    float pring[3][16];
    auto issue_p = [&](float (&t16)[16], int jt) __attribute__((always_inline)) {
      const int jj = jt * 32 + l31;
      const unsigned coff = (jj < T ? (unsigned)jj * 4u : 0x80000000u) + (unsigned)(it * 32 + h) * lp.T4;
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) t16[k2] = lp(coff, 2 * k2);
    };
    issue_p(pring[0], 0);
    issue_p(pring[1], 1);
    static_for<0, NKT>([&](auto JT) {
      constexpr int jt = decltype(JT)::value;
      const bool near = jt + 1 >= it && jt <= it + 1;
      const int r1 = jt * 32 + 4 * h - i + w + 1;
      float rv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) rv[e] = 0.f;
      if (near) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned idx = (unsigned)(r1 + (e & 3) + 8 * (e >> 2));
          rv[e] = relw[l31 * RWP + (idx < (unsigned)(RWP - 1) ? idx : (unsigned)(RWP - 1))];
        }
      }
      if (jt + 2 < NKT) issue_p(pring[(jt + 2) % 3], jt + 2);
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) ptile[(2 * k2 + h) * PTP + l31] = pring[jt % 3][k2];
      float cd[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) cd[e] = 1.f;
      if (p.pdrop > 0.f) {
#pragma unroll
        for (int e = 0; e < 16; ++e) cd[e] = drop_scale(p.seed + (p.seed_off ? *p.seed_off : 0ull), rowoff + jt * 32 + (e & 3) + 8 * (e >> 2), p.pdrop, inv_keep);
      }
      unsigned wl = 0;
      if constexpr (masked) {
        wl = (unsigned)(mb[jt >> 1] >> ((jt & 1) * 32)) >> (4 * h);
        wl = mi != 0.f ? wl : 0u;
      }
      float pd[16], dsv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = ptile[l31 * PTP + (e & 3) + 8 * (e >> 2) + 4 * h];
        const float dpc = (acc[jt][e] + rv[e]) * cd[e];
        float ds = pv * (dpc - dot);
        if constexpr (masked) ds = (wl & (1u << ((e & 3) + 8 * (e >> 2)))) ? ds : 0.f;
        pd[e] = pv * cd[e];
        dsv[e] = ds;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        acc[jt][e] = dsv[e];
        ptile[l31 * PTP + (e & 3) + 8 * (e >> 2) + 4 * h] = dsv[e];
      }
      const int jj = jt * 32 + l31;
      float* drow = p.dS + ((size_t)g * T + it * 32 + h) * T + jj;
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) {
        const float v = ptile[(2 * k2 + h) * PTP + l31];
        if (it * 32 + 2 * k2 + h < T && jj < T) drow[(size_t)(2 * k2) * T] = v;
      }
      if (near) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const unsigned idx = (unsigned)(r1 + (e & 3) + 8 * (e >> 2));
          const unsigned ic = idx < (unsigned)(RWP - 1) ? idx : (unsigned)(RWP - 1);
          relp[l31 * RWP + ic] = pd[e];
          rels[l31 * RWP + ic] = dsv[e];
        }
      }
    });

    VCV_STAMP(4);
    // ---- dQ^T = qscale (K dS^T + embk^T band(dS))
    f32x16 oacc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[t][e] = 0.f;
    const float* Kr = Xs + l31 * TPV + 4 * h;
    static_for<0, NKT>([&](auto JT) {
      constexpr int jt = decltype(JT)::value;
      if (jt < nkt) {
        if constexpr (BF) {
#pragma unroll
          for (int m2 = 0; m2 < 2; ++m2) {
            bf16x8 bfr;
#pragma unroll
            for (int e = 0; e < 8; ++e) bfr[e] = (__bf16)acc[jt][8 * m2 + e];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              bf16x8 afr;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const int s2 = 8 * m2 + e;
                afr[e] = (__bf16)Kr[32 * t * TPV + jt * 32 + (s2 & 3) + 8 * (s2 >> 2)];
              }
              oacc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr, bfr, oacc[t], 0, 0, 0);
            }
          }
        } else {
#pragma unroll
          for (int s2 = 0; s2 < 16; ++s2) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
              oacc[t] = mma32(Kr[32 * t * TPV + jt * 32 + (s2 & 3) + 8 * (s2 >> 2)], acc[jt][s2], oacc[t]);
          }
        }
      }
    });
#pragma unroll
    for (int s = 0; s < 8; ++s)
      if (2 * s < nr) {
        const float bfr = rels[l31 * RWP + 1 + 2 * s + h];
#pragma unroll
        for (int t = 0; t < NT; ++t) oacc[t] = mma32(etab[(2 * s + h) * DK + 32 * t + l31], bfr, oacc[t]);
      }
    if (i < T) {
      float* og = p.dq + (size_t)g * DK * T + i;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) og[(size_t)(32 * t + acc_row(e, h)) * T] = oacc[t][e] * p.qscale;
    }

    VCV_STAMP(5);
    // ---- table gradients of this query tile: D[r][d] = sum_i band[i][r] X[d][i] (X = Q with band(dS), dO with band(Pd)):
    // A = the band scratch read with lane = r, B = X rows d at the wave's queries (row-strided loads, once per wave)
    // Out as this tile's PARTIAL table (workspace behind dS: [head][query tile][dembk | dembv][r][d]); rel_attn_demb_reduce_kernel
    // adds the partials in a fixed order (the 896 waves of a B = 32 launch adding into the same 1,152 words with atomics
    // took 20 us of this kernel's 78, and left the two gradients order-dependent)
    auto table_grad = [&](const float* band, const ColLoader& lx, float* dst, float scale) __attribute__((always_inline)) {
      f32x16 dacc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) dacc[t][e] = 0.f;
      const int rl = l31 < RWP - 2 ? l31 : RWP - 2;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const float a = band[(2 * s + h) * RWP + 1 + rl];
        const int ii = it * 32 + 2 * s + h;
        const unsigned coff = ii < T ? (unsigned)ii * 4u : 0x80000000u;
#pragma unroll
        for (int t = 0; t < NT; ++t) dacc[t] = mma32(a, lx(coff + (unsigned)l31 * lx.T4, 32 * t), dacc[t]);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int r = acc_row(e, h);
          if (r < nr) dst[r * DK + 32 * t + l31] = dacc[t][e] * scale;
        }
    };
    float* part = p.dS + (size_t)p.B * p.H * T * T + ((size_t)g * nkt + it) * 2 * nr * DK;
    table_grad(rels, lq, part, p.qscale);
    table_grad(relp, lo, part + nr * DK, 1.f);
    VCV_STAMP(6);
    VCV_STAMP(7);
  };
  if (allone && __ballot(mi == 0.f) == 0ull) rest(std::false_type());
  else rest(std::true_type());
}

template <bool BF, int NKT, int DK>
__global__ void __launch_bounds__(64 * FW) rel_attn_bwd_cols2_kernel(const AttnArgs p) {
  constexpr int NT = DK / 32, TPV = NKT * 32 + 1;
  static_assert(NKT * 32 == 64 * FW, "one staged column per thread");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = p.T;
  const int nkt = (T + 31) >> 5;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* Os = sm;               // dO image [DK][TPV]
  float* Qs = Os + DK * TPV;    // Q image
  const int g = blockIdx.y;
  const int jt = blockIdx.x * FW + wave;  // this wave's key tile
  const ColLoader lq = col_loader(p.q + (size_t)g * DK * T, DK, T), lo = col_loader(p.dO + (size_t)g * DK * T, DK, T);
  const ColLoader lp = col_loader(p.Pin + (size_t)g * T * T, T, T), ls = col_loader(p.dS + (size_t)g * T * T, T, T);
  {
    const unsigned coff = tid < T ? (unsigned)tid * 4u : 0x80000000u;
    float stg[DK];
#pragma unroll
    for (int u = 0; u < DK; ++u) stg[u] = lo(coff, u);
#pragma unroll
    for (int u = 0; u < DK; ++u) Os[u * TPV + tid] = stg[u];
#pragma unroll
    for (int u = 0; u < DK; ++u) stg[u] = lq(coff, u);
#pragma unroll
    for (int u = 0; u < DK; ++u) Qs[u * TPV + tid] = stg[u];
  }
  __syncthreads();
  if (jt >= nkt) return;
  const int j = jt * 32 + l31;
  const unsigned joff = j < T ? (unsigned)j * 4u : 0x80000000u;
  const float inv_keep = p.pdrop > 0.f ? 1.f / (1.f - p.pdrop) : 1.f;
  f32x16 av[NT], ak[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) av[t][e] = ak[t][e] = 0.f;
  const float* Ol = Os + l31 * TPV + 4 * h;
  const float* Ql = Qs + l31 * TPV + 4 * h;
  float pv[2][16], ds[2][16];
  auto load_tiles = [&](float (&pq)[16], float (&dq_)[16], int it) __attribute__((always_inline)) {
    const unsigned roff = joff + (unsigned)(it * 32 + 4 * h) * lp.T4;  // (a row past T lands past the matrix: 0)
#pragma unroll
    for (int e = 0; e < 16; ++e) pq[e] = lp(roff, (e & 3) + 8 * (e >> 2));
#pragma unroll
    for (int e = 0; e < 16; ++e) dq_[e] = ls(roff, (e & 3) + 8 * (e >> 2));
  };
  load_tiles(pv[0], ds[0], 0);
  static_for<0, NKT>([&](auto IT) {
    constexpr int it = decltype(IT)::value;
    if (it < nkt) {
      if (it + 1 < nkt) load_tiles(pv[(it + 1) & 1], ds[(it + 1) & 1], it + 1);
      float (&pc)[16] = pv[it & 1];
      float (&dc)[16] = ds[it & 1];
      if (p.pdrop > 0.f) {
#pragma unroll
        for (int e = 0; e < 16; ++e)
          pc[e] *= drop_scale(p.seed + (p.seed_off ? *p.seed_off : 0ull), ((size_t)g * T + it * 32 + 4 * h + (e & 3) + 8 * (e >> 2)) * T + j, p.pdrop, inv_keep);
      }
      if constexpr (BF) {
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2) {
          bf16x8 bp, bd;
#pragma unroll
          for (int e = 0; e < 8; ++e) bp[e] = (__bf16)pc[8 * m2 + e], bd[e] = (__bf16)dc[8 * m2 + e];
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            bf16x8 ao, aq;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int s2 = 8 * m2 + e, off = 32 * t * TPV + it * 32 + (s2 & 3) + 8 * (s2 >> 2);
              ao[e] = (__bf16)Ol[off], aq[e] = (__bf16)Ql[off];
            }
            av[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ao, bp, av[t], 0, 0, 0);
            ak[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aq, bd, ak[t], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            const int off = 32 * t * TPV + it * 32 + (s2 & 3) + 8 * (s2 >> 2);
            av[t] = mma32(Ol[off], pc[s2], av[t]);
            ak[t] = mma32(Ql[off], dc[s2], ak[t]);
          }
        }
      }
    }
  });
  if (j < T) {
    float* dvp = p.dv + (size_t)g * DK * T + j;
    float* dkp = p.dk_ + (size_t)g * DK * T + j;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const size_t ro = (size_t)(32 * t + acc_row(e, h)) * T;
        dvp[ro] = av[t][e];
        dkp[ro] = ak[t][e] * p.qscale;
      }
  }
}

// dembk / dembv [nr * dk] = sum over the (head, query tile) partial tables, in index order (deterministic).  Block = 64 table
// elements x 16 slices of the partial range; the slices meet in LDS.
__global__ void __launch_bounds__(1024) rel_attn_demb_reduce_kernel(const float* __restrict__ part, float* __restrict__ dembk,
                                                                   float* __restrict__ dembv, int npart, int ne) {
  __shared__ float red[16][65];
  const int x = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int el = blockIdx.x * 64 + x;  // element of the concatenated [dembk | dembv] table (2 * ne floats per partial)
  float a = 0.f;
  if (el < 2 * ne) {
    const int per = (npart + 15) / 16;
    const int p0 = sl * per, p1 = p0 + per < npart ? p0 + per : npart;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int q = p0;
    for (; q + 3 < p1; q += 4) {
      a0 += part[(size_t)q * 2 * ne + el];
      a1 += part[(size_t)(q + 1) * 2 * ne + el];
      a2 += part[(size_t)(q + 2) * 2 * ne + el];
      a3 += part[(size_t)(q + 3) * 2 * ne + el];
    }
    for (; q < p1; ++q) a0 += part[(size_t)q * 2 * ne + el];
    a = (a0 + a1) + (a2 + a3);
  }
  red[sl][x] = a;
  __syncthreads();
  if (sl == 0 && el < 2 * ne) {
    float t = red[0][x];
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) t += red[k2][x];
    if (el < ne) dembk[el] = t; else dembv[el - ne] = t;
  }
}

size_t lds_bwd_rows2(int nkt_max, int dk) {
  return sizeof(float) * ((size_t)dk * (nkt_max * 32 + 1) + RELP * dk + FW * (3 * 32 * RWP + 32 * PTP));
}
size_t lds_bwd_cols2(int nkt_max, int dk) { return sizeof(float) * ((size_t)2 * dk * (nkt_max * 32 + 1)); }

size_t lds_rows(int nkt_max, int dk) {
  return sizeof(float) * ((size_t)nkt_max * 32 + (size_t)2 * dk * (nkt_max * 32 + 1) + RELP * dk + FW * (32 * RWP + 32 * PTP));
}

// (the BSF floats hold the band-logit staging (64 * QP + RELP * 64) and the partial-output tiles (NWV * 32 * OP) too)
size_t lds_bytes(int TP) { return sizeof(float) * ((size_t)32 * TP + 32 * RELP + TP + RELP * 64 + BSF); }

bool ok_shape(int B, int H, int dk, int T, int w) {
  return B > 0 && H > 0 && dk > 0 && dk <= 64 && (dk % 2) == 0 && T > 0 && w >= 0 && 2 * w + 1 <= RELP &&
         lds_bytes(((T + 63) & ~63) + 2) <= VCV_LDS_LIMIT;  // the 32-row probability tile must fit: T <= 896
}

}  // namespace

// 0: this shape runs on the fused kernels (dk <= 64 even, T <= 896, window <= 7); else the caller keeps the unfused path
extern "C" int vcv_rel_attn_supported(int B, int H, int dk, int T, int w) { return ok_shape(B, H, dk, T, w) ? 0 : VCV_EINVAL; }

// out [B, H*dk, T] = attention(q, k, v) with relative keys / values `embk`, `embv` [2w+1, dk], mask [B, T].
// P / Pd [B*H, T, T] (either may be null): softmax probabilities before / after dropout.
extern "C" int vcv_rel_attn_fwd(const float* q, const float* k, const float* v, const float* embk, const float* embv,
                                const float* mask, float* out, float* P, float* Pd, int B, int H, int dk, int T, int w,
                                float qscale, float pdrop, uint64_t seed, int bf16, void* stream) {
  if (!q || !k || !v || !embk || !embv || !mask || !out || !ok_shape(B, H, dk, T, w) || pdrop < 0.f || pdrop >= 1.f) return VCV_EINVAL;
  AttnArgs a = {};
  a.q = q, a.k = k, a.v = v, a.embk = embk, a.embv = embv, a.mask = mask, a.out = out, a.P = P, a.Pd = Pd;
  a.B = B, a.H = H, a.dk = dk, a.T = T, a.w = w, a.TP = ((T + 63) & ~63) + 2, a.qscale = qscale, a.pdrop = pdrop, a.seed = seed;
  a.seed_off = (const unsigned long long*)vcv_get_seed_offset_ptr();
#ifdef VCV_ATTN_STAMPS
  if (!Pd) a.Pd = (float*)g_attn_stamps;
#endif
  // T <= 512: one wave per 32 query rows, scores in registers (rel_attn_fwd_rows_kernel); VCVITS_ATTN_ROWS=0 keeps the
  // workgroup-per-tile kernel
  const bool rows_on = vcv_tuning().attn_rows != 0;
  // (T <= 256: the 8 x 16 score registers of a query; a 16-tile instance spills ~500 registers and measured slower than the
  // workgroup-per-tile kernel at T = 500, so longer sequences stay there)
  if (rows_on && T <= 256 && (dk == 32 || dk == 64) && 2 * w + 1 <= RELP) {
    const int nkt_max = 8;
    const size_t ldsr = lds_rows(nkt_max, dk);
    void (*kr)(const AttnArgs);
#define VCV_ATTN_ROWS_PICK(BFV) kr = dk == 64 ? rel_attn_fwd_rows_kernel<BFV, 8, 64> : rel_attn_fwd_rows_kernel<BFV, 8, 32>
    if (bf16) { VCV_ATTN_ROWS_PICK(true); } else { VCV_ATTN_ROWS_PICK(false); }
#undef VCV_ATTN_ROWS_PICK
    if (ldsr > 64 * 1024 && hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr) != hipSuccess)
      return VCV_EHIP;
    const double flops = 4.0 * B * H * (double)T * T * dk;
    const int tag[12] = {B, bf16 ? 2 : 4, dk, H, 0, T, 1, 1, 1, 100, 32 * 1000 + nkt_max * 32, 1};
    hipEvent_t ev0, ev1;
    vcv_prof_events(VCV_PROF_ATTN, flops, tag, 12, &ev0, &ev1, 0.0, bf16 ? flops / VCV_PEAK_BF16_MFMA : 0.0);
    const int nqt = (T + 31) / 32;
    VCV_LAUNCH_EV(kr, dim3((nqt + FW - 1) / FW, B * H), dim3(64 * FW), (unsigned)ldsr, (hipStream_t)stream, ev0, ev1, a);
    return vcv_check_launch();
  }
  const size_t lds = lds_bytes(a.TP);
  auto kern = bf16 ? rel_attn_fwd_kernel<true> : rel_attn_fwd_kernel<false>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VCV_EHIP;
  const double flops = 4.0 * B * H * (double)T * T * dk;
  const int tag[12] = {B, bf16 ? 2 : 4, dk, H, 0, T, 1, 1, 1, 100, 32 * 1000 + 32, 0};
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_ATTN, flops, tag, 12, &ev0, &ev1, 0.0, bf16 ? flops / VCV_PEAK_BF16_MFMA : 0.0);
  VCV_LAUNCH_EV(kern, dim3((T + 31) / 32, B * H), dim3(NTH), (unsigned)lds, (hipStream_t)stream, ev0, ev1, a);
  return vcv_check_launch();
}

// Gradients of vcv_rel_attn_fwd.  P: the probabilities the forward saved; dS: [B*H, T, T] workspace; dembk / dembv
// [2w+1, dk] are overwritten.
// vcv_rel_attn_bwd2: the same with the forward's output `out` [B, H*dk, T] (may be null); dS is then a workspace of
// B*H*T*T + B*H*ceil(T/32)*2*(2w+1)*dk floats (the per-tile partial tables of the two table gradients follow dS).  With it the row pass knows
// sum_j dPd Pd of every query (= sum_d dO out) before its first tile and forms dS in one pass (T <= 256, dk 32 / 64:
// rel_attn_bwd_rows2_kernel / rel_attn_bwd_cols2_kernel); without it, or for other shapes, the workgroup-per-tile kernels run.
extern "C" int vcv_rel_attn_bwd2(const float* q, const float* k, const float* v, const float* embk, const float* embv,
                                 const float* mask, const float* P, const float* out, const float* dO, float* dS, float* dq,
                                 float* dk_out, float* dv, float* dembk, float* dembv, int B, int H, int dk, int T, int w,
                                 float qscale, float pdrop, uint64_t seed, int bf16, void* stream) {
  if (!q || !k || !v || !embk || !embv || !mask || !P || !dO || !dS || !dq || !dk_out || !dv || !dembk || !dembv ||
      !ok_shape(B, H, dk, T, w) || pdrop < 0.f || pdrop >= 1.f)
    return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  AttnArgs a = {};
  a.q = q, a.k = k, a.v = v, a.embk = embk, a.embv = embv, a.mask = mask, a.Pin = P, a.dO = dO, a.out = (float*)out;
  a.seed_off = (const unsigned long long*)vcv_get_seed_offset_ptr();  // (a captured training pass: the same offset as its forward)
#ifdef VCV_ATTN_STAMPS
  a.Pd = (float*)g_attn_stamps;
#endif
  a.dS = dS, a.dq = dq, a.dk_ = dk_out, a.dv = dv, a.dembk = dembk, a.dembv = dembv;
  a.B = B, a.H = H, a.dk = dk, a.T = T, a.w = w, a.TP = ((T + 63) & ~63) + 2, a.qscale = qscale, a.pdrop = pdrop, a.seed = seed;
  const bool rows_on = vcv_tuning().attn_rows != 0;
  if (rows_on && out && T <= 256 && (dk == 32 || dk == 64) && 2 * w + 1 <= RELP) {
    void (*kr)(const AttnArgs);
    void (*kc)(const AttnArgs);
    if (bf16) {
      kr = dk == 64 ? rel_attn_bwd_rows2_kernel<true, 8, 64> : rel_attn_bwd_rows2_kernel<true, 8, 32>;
      kc = dk == 64 ? rel_attn_bwd_cols2_kernel<true, 8, 64> : rel_attn_bwd_cols2_kernel<true, 8, 32>;
    } else {
      kr = dk == 64 ? rel_attn_bwd_rows2_kernel<false, 8, 64> : rel_attn_bwd_rows2_kernel<false, 8, 32>;
      kc = dk == 64 ? rel_attn_bwd_cols2_kernel<false, 8, 64> : rel_attn_bwd_cols2_kernel<false, 8, 32>;
    }
    const size_t l1 = lds_bwd_rows2(8, dk), l2 = lds_bwd_cols2(8, dk);
    if ((l1 > 64 * 1024 && hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l1) != hipSuccess) ||
        (l2 > 64 * 1024 && hipFuncSetAttribute((const void*)kc, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2) != hipSuccess))
      return VCV_EHIP;
    const double flops = 4.0 * B * H * (double)T * T * dk;
    const int nqt = (T + 31) / 32;
    const dim3 grid((nqt + FW - 1) / FW, B * H), block(64 * FW);
    hipEvent_t ev0, ev1;
    const int tag[12] = {B, bf16 ? 2 : 4, dk, H, 0, T, 1, 1, 1, 101, 32 * 1000 + 256, 1};
    vcv_prof_events(VCV_PROF_ATTN, flops, tag, 12, &ev0, &ev1, 0.0, bf16 ? flops / VCV_PEAK_BF16_MFMA : 0.0);
    VCV_LAUNCH_EV(kr, grid, block, (unsigned)l1, st, ev0, ev1, a);
    const int tag2[12] = {B, bf16 ? 2 : 4, dk, H, 0, T, 1, 1, 1, 102, 32 * 1000 + 256, 1};
    vcv_prof_events(VCV_PROF_ATTN, flops, tag2, 12, &ev0, &ev1, 0.0, bf16 ? flops / VCV_PEAK_BF16_MFMA : 0.0);
    VCV_LAUNCH_EV(kc, grid, block, (unsigned)l2, st, ev0, ev1, a);
    const int nel = (2 * w + 1) * dk;
    hipLaunchKernelGGL(rel_attn_demb_reduce_kernel, dim3((2 * nel + 63) / 64), dim3(1024), 0, st,
                       (const float*)(dS + (size_t)B * H * T * T), dembk, dembv, B * H * nqt, nel);
    return vcv_check_launch();
  }
  const size_t ne = sizeof(float) * (2 * w + 1) * dk;
  if (vcv_zero_async(dembk, ne, st) != hipSuccess || vcv_zero_async(dembv, ne, st) != hipSuccess) return VCV_EHIP;
  const size_t lds = lds_bytes(a.TP);
  auto rows = bf16 ? rel_attn_bwd_rows_kernel<true> : rel_attn_bwd_rows_kernel<false>;
  auto cols = bf16 ? rel_attn_bwd_cols_kernel<true> : rel_attn_bwd_cols_kernel<false>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VCV_EHIP;
  const double flops = 4.0 * B * H * (double)T * T * dk;
  const int tag[12] = {B, bf16 ? 2 : 4, dk, H, 0, T, 1, 1, 1, 101, 32 * 1000 + 32, 0};
  hipEvent_t ev0, ev1;
  vcv_prof_events(VCV_PROF_ATTN, flops, tag, 12, &ev0, &ev1, 0.0, bf16 ? flops / VCV_PEAK_BF16_MFMA : 0.0);
  VCV_LAUNCH_EV(rows, dim3((T + 31) / 32, B * H), dim3(NTH), (unsigned)lds, st, ev0, ev1, a);
  const int tag2[12] = {B, bf16 ? 2 : 4, dk, H, 0, T, 1, 1, 1, 102, 32 * 1000 + 32, 0};
  vcv_prof_events(VCV_PROF_ATTN, flops, tag2, 12, &ev0, &ev1, 0.0, bf16 ? flops / VCV_PEAK_BF16_MFMA : 0.0);
  const size_t lds2 = sizeof(float) * 2 * BSF;
  if (lds2 > 64 * 1024 && hipFuncSetAttribute((const void*)cols, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess)
    return VCV_EHIP;
  VCV_LAUNCH_EV(cols, dim3((T + 31) / 32, B * H), dim3(NTH), (unsigned)lds2, st, ev0, ev1, a);
  return vcv_check_launch();
}

extern "C" int vcv_rel_attn_bwd(const float* q, const float* k, const float* v, const float* embk, const float* embv,
                                const float* mask, const float* P, const float* dO, float* dS, float* dq, float* dk_out,
                                float* dv, float* dembk, float* dembv, int B, int H, int dk, int T, int w, float qscale,
                                float pdrop, uint64_t seed, int bf16, void* stream) {
  return vcv_rel_attn_bwd2(q, k, v, embk, embv, mask, P, nullptr, dO, dS, dq, dk_out, dv, dembk, dembv, B, H, dk, T, w, qscale,
                           pdrop, seed, bf16, stream);
}

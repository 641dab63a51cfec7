// conv_pack.hip -- batched weight packs: every packed-weight buffer of a module tree in ONE launch.
//
// The packed-operand conv kernels (conv_pk.hip: fp32 / bf16 elements; conv_x3.hip: three bf16 terms per weight) read their
// weights from buffers packed in LDS image order.  A pack is valid until the weights change, i.e. for one optimizer step:
// packing lazily, one launch per (layer, layout), was 180-270 launches of ~10 us per step (2.3 % of the fp32 step, 6.7 %
// of the bf16 step).  The host records each pack it had to make as a job (vcv_conv_*_pack_job fills the layout from the
// launch's own plan) and, the next time the tree's weights are re-normalised, replays all of them here.
// The per-element bodies restate pack_pk_kernel (conv_pk.hip) and pack_x3_kernel (conv_x3.hip); tests/test_pack_many_gpu.py
// compares the buffers bit for bit with the ones those kernels write.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float pack_src(const VcvPackJob& J, int r, int m, int c, int j) {
  if (m >= J.M || c >= J.C) return 0.f;
  if (J.mode == 0) return j < J.K ? J.w[((size_t)m * J.C + c) * J.K + j] : 0.f;
  if (J.mode == 1) return j < J.K ? J.w[((size_t)c * J.M + m) * J.K + (J.K - 1 - j)] : 0.f;
  const int k = r + j * J.phases;
  return k < J.K ? J.w[((size_t)c * J.M + m) * J.K + k] : 0.f;
}

// kind 0: conv_x3 (wp[phase][m-tile][group][j][term][h][m][8] bf16, three planes; one thread per (.., h, m))
__device__ __forceinline__ void pack_x3(const VcvPackJob& J, size_t i) {
  size_t t = i;
  const int ml = (int)(t % J.BM); t /= J.BM;
  const int hh = (int)(t & 1); t >>= 1;
  const int j = (int)(t % J.JA); t /= J.JA;
  const int g = (int)(t % J.nch); t /= J.nch;
  const int mt = (int)(t % J.nmt); t /= J.nmt;
  const int r = (int)t;
  const int m = mt * J.BM + ml;
  const int c0 = g * 16 + hh * 8;
  bf16x8 v0, v1, v2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float f = pack_src(J, r, m, c0 + e, j);
    const __bf16 a = (__bf16)f;
    const float r1 = f - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 d = (__bf16)(r1 - (float)b);
    v0[e] = a, v1[e] = b, v2[e] = d;
  }
  const size_t slab = (((size_t)r * J.nmt + mt) * J.nch + g) * J.JA + j;
  bf16x8* o = (bf16x8*)J.wp + slab * (size_t)(3 * 2 * J.BM) + (size_t)hh * J.BM + ml;
  o[0] = v0;
  o[(size_t)2 * J.BM] = v1;
  o[(size_t)4 * J.BM] = v2;
}

// kind 1 / 2: conv_pk fp32 (4 channels per 16-byte group) / bf16 (8): wp[phase][m-tile][chunk][j][cg][h][m][group]
template <int CPG>
__device__ __forceinline__ void pack_pk(const VcvPackJob& J, size_t i) {
  const int ncg = J.BKC / (2 * CPG);
  size_t t = i;
  const int ml = (int)(t % J.BM); t /= J.BM;
  const int hh = (int)(t & 1); t >>= 1;
  const int cg = (int)(t % ncg); t /= ncg;
  const int j = (int)(t % J.JA); t /= J.JA;
  const int ch = (int)(t % J.nch); t /= J.nch;
  const int mt = (int)(t % J.nmt); t /= J.nmt;
  const int r = (int)t;
  const int m = mt * J.BM + ml;
  const int c0 = ch * J.BKC + cg * 2 * CPG + hh * CPG;
  if (CPG == 4) {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = pack_src(J, r, m, c0 + e, j);
    ((f32x4*)J.wp)[i] = v;
  } else {
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (__bf16)pack_src(J, r, m, c0 + e, j);
    ((bf16x8*)J.wp)[i] = v;
  }
}

// kind 0, coalesced: one workgroup per (phase, m-tile, 16-channel group, 32-row slice).  The slice's source weights are 32
// (mode 0: [m][16*K]) or 16 (modes 1 / 2: [c][32*K]) contiguous runs -- read coalesced into a ~10 KB LDS tile (16 workgroups
// per CU: the loads of one hide behind the gathers of the others), packed planes gathered from there.  The
// thread-per-item body above reads its 8 channels K floats apart and rows C*K floats apart across lanes: every lane-load
// its own cache line.  Same bits as pack_x3 (tests/test_pack_many_gpu.py).
constexpr int PK_RT = 32;

__global__ void __launch_bounds__(256) pack_x3_tile_kernel(const VcvPackJob* __restrict__ jobs, int n) {
  extern __shared__ float src[];
  int lo = 0, hi = n - 1;
  const long long bx = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= bx) lo = mid; else hi = mid - 1;
  }
  const VcvPackJob J = jobs[lo];
  const int nsl = J.BM / PK_RT;  // row slices per m-tile
  int t = (int)(bx - J.block0);
  const int sl = t % nsl; t /= nsl;
  // kind 0: one 16-channel group g of the split layout.  kind 2 (bf16 elements of conv_pk.hip, round 4: the thread-per-item
  // body ran the bf16 steps' packs at 0.65 TB/s, 1.6-1.8 ms per step): 16-channel pair `cg` of reduction chunk `ch`
  int g, cg = 0, ch = 0;
  const int ncg = J.kind == 2 ? J.BKC / 16 : 1;
  if (J.kind == 2) {
    cg = t % ncg; t /= ncg;
    ch = t % J.nch; t /= J.nch;
    g = 0;
  } else {
    g = t % J.nch; t /= J.nch;
  }
  const int mt = t % J.nmt;
  const int r = t / J.nmt;
  const int tid = threadIdx.x;
  const int ml0 = sl * PK_RT, m0 = mt * J.BM + ml0, c0 = J.kind == 2 ? ch * J.BKC + cg * 16 : g * 16;
  const int K = J.K;
  int rows, L;
  if (J.mode == 0) rows = PK_RT, L = 16 * K; else rows = 16, L = PK_RT * K;
  const int pitch = L | 1;
  // valid rows / valid part of a row (the rest of the tile reads as zero)
  int vrows = J.mode == 0 ? J.M - m0 : J.C - c0;
  vrows = vrows < 0 ? 0 : (vrows < rows ? vrows : rows);
  int vL = J.mode == 0 ? (J.C - c0) * K : (J.M - m0) * K;
  vL = vL < 0 ? 0 : (vL < L ? vL : L);
  const size_t rstride = J.mode == 0 ? (size_t)J.C * K : (size_t)J.M * K;
  const float* base = J.mode == 0 ? J.w + ((size_t)m0 * J.C + c0) * K : J.w + ((size_t)c0 * J.M + m0) * K;
  const int tot = rows * L;
  for (int i0 = tid; i0 < tot; i0 += 256 * 6) {
    float v[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = i0 + u * 256;
      const int row = i / L, col = i - row * L;
      v[u] = (i < tot && row < vrows && col < vL) ? base[(size_t)row * rstride + col] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const int i = i0 + u * 256;
      if (i < tot) {
        const int row = i / L, col = i - row * L;
        src[row * pitch + col] = v[u];
      }
    }
  }
  __syncthreads();
  const int items = J.JA * 2 * PK_RT;
  if (J.kind == 2) {
    // wp[phase][m-tile][chunk][j][cg][h][m][8 bf16]: one 16-byte item per (j, h, row)
    const size_t chunk0 = (((size_t)r * J.nmt + mt) * J.nch + ch) * J.JA;
    for (int it = tid; it < items; it += 256) {
      const int mr = it % PK_RT;
      const int hh = (it / PK_RT) & 1;
      const int j = it / (2 * PK_RT);
      const int kk = J.mode == 0 ? j : J.mode == 1 ? K - 1 - j : r + j * J.phases;
      const bool kok = J.mode == 2 ? kk < K : j < K;
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float f = 0.f;
        if (kok) f = J.mode == 0 ? src[mr * pitch + (hh * 8 + e) * K + kk] : src[(hh * 8 + e) * pitch + mr * K + kk];
        v[e] = (__bf16)f;
      }
      ((bf16x8*)J.wp)[((((chunk0 + j) * ncg + cg) * 2 + hh) * (size_t)J.BM) + ml0 + mr] = v;
    }
    return;
  }
  const size_t slab0 = (((size_t)r * J.nmt + mt) * J.nch + g) * J.JA;
  for (int it = tid; it < items; it += 256) {
    const int mr = it % PK_RT;
    const int hh = (it / PK_RT) & 1;
    const int j = it / (2 * PK_RT);
    const int kk = J.mode == 0 ? j : J.mode == 1 ? K - 1 - j : r + j * J.phases;
    const bool kok = J.mode == 2 ? kk < K : j < K;
    bf16x8 v0, v1, v2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float f = 0.f;
      if (kok) f = J.mode == 0 ? src[mr * pitch + (hh * 8 + e) * K + kk] : src[(hh * 8 + e) * pitch + mr * K + kk];
      const __bf16 a = (__bf16)f;
      const float r1 = f - (float)a;
      const __bf16 b = (__bf16)r1;
      const __bf16 d = (__bf16)(r1 - (float)b);
      v0[e] = a, v1[e] = b, v2[e] = d;
    }
    bf16x8* o = (bf16x8*)J.wp + (slab0 + j) * (size_t)(3 * 2 * J.BM) + (size_t)hh * J.BM + ml0 + mr;
    o[0] = v0;
    o[(size_t)2 * J.BM] = v1;
    o[(size_t)4 * J.BM] = v2;
  }
}

__global__ void __launch_bounds__(256) pack_many_kernel(const VcvPackJob* __restrict__ jobs, int n) {
  // the job of this block: the last one whose first block is <= blockIdx.x (binary search; jobs are sorted by block0)
  int lo = 0, hi = n - 1;
  const long long bx = blockIdx.x;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= bx) lo = mid; else hi = mid - 1;
  }
  const VcvPackJob J = jobs[lo];
  const size_t i = (size_t)(bx - J.block0) * 256 + threadIdx.x;
  if (i >= (size_t)J.total) return;
  if (J.kind == 0) pack_x3(J, i);
  else if (J.kind == 1) pack_pk<4>(J, i);
  else pack_pk<8>(J, i);
}

}  // namespace

// jobs: host array of n jobs with w / wp / the layout fields set (vcv_conv_*_pack_job) -- block0 is filled here;
// table_dev: device scratch of n * sizeof(VcvPackJob) bytes.  One launch packs them all.
static int pack_many_impl(VcvPackJob* jobs, int n, void* table_dev, void* stream, bool upload) {
  if (!jobs || n <= 0 || !table_dev) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const bool no_tile = !vcv_tuning().pack_tile;
  constexpr size_t TILE_LDS_MAX = 40 * 1024;
  auto tile_lds = [](const VcvPackJob& j) {
    return sizeof(float) * (size_t)(j.mode == 0 ? PK_RT * ((16 * j.K) | 1) : 16 * ((PK_RT * j.K) | 1));
  };
  // the split-operand jobs go first (coalesced tile kernel: one workgroup per 32-row slice of a 16-channel group), the
  // thread-per-item jobs after them: two launches over the two parts of one table (the host array is re-ordered in place)
  int nt = 0;
  for (int i = 0; i < n; ++i) {
    VcvPackJob& j = jobs[i];
    if (!j.w || !j.wp || j.total <= 0 || j.kind < 0 || j.kind > 2) return VCV_EINVAL;
    const bool no_tile2 = !vcv_tuning().pack_tile_bf16;
    const bool tile = (j.kind == 0 || (j.kind == 2 && !no_tile2 && j.BKC % 16 == 0 && j.mode <= 2)) && !no_tile &&
                      j.BM % PK_RT == 0 && tile_lds(j) <= TILE_LDS_MAX && j.total % ((int64_t)j.JA * 2 * j.BM) == 0;
    j.reserved = tile ? 1 : 0;
    if (tile) {
      const VcvPackJob tmp = jobs[nt];
      jobs[nt] = j;
      jobs[i] = tmp;
      ++nt;
    }
  }
  long long blocks_t = 0, blocks = 0;
  size_t lds_max = 0;
  for (int i = 0; i < nt; ++i) {
    const VcvPackJob& j = jobs[i];
    jobs[i].block0 = blocks_t;
    blocks_t += j.total / ((int64_t)j.JA * 2 * j.BM) * (j.BM / PK_RT);
    if (tile_lds(j) > lds_max) lds_max = tile_lds(j);
  }
  for (int i = nt; i < n; ++i) {
    jobs[i].block0 = blocks;
    blocks += (jobs[i].total + 255) / 256;
  }
  if (blocks >= (1ll << 31) || blocks_t >= (1ll << 31)) return VCV_EINVAL;
  if (upload && hipMemcpyAsync(table_dev, jobs, sizeof(VcvPackJob) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) return VCV_EHIP;
  if (nt > 0)
    hipLaunchKernelGGL(pack_x3_tile_kernel, dim3((unsigned)blocks_t), dim3(256), lds_max, st, (const VcvPackJob*)table_dev, nt);
  if (n > nt)
    hipLaunchKernelGGL(pack_many_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const VcvPackJob*)table_dev + nt, n - nt);
  return vcv_check_launch();
}

extern "C" int vcv_pack_many(VcvPackJob* jobs, int n, void* table_dev, void* stream) {
  return pack_many_impl(jobs, n, table_dev, stream, true);
}

// The same launches WITHOUT the host -> device copy of the job table: `jobs` is finalised in place (order, block0) and the
// caller uploads it into `table_dev` itself before the launches first EXECUTE.  For launch sequences recorded into a HIP
// graph: the table's contents never change between replays, so the graph holds no copy node that re-reads host memory
// (light/graphed.py uploads every table of a captured pass once, eagerly, right after the capture).
extern "C" int vcv_pack_many_prepared(VcvPackJob* jobs, int n, void* table_dev, void* stream) {
  return pack_many_impl(jobs, n, table_dev, stream, false);
}

// Host table -> device on `stream` (hipMemcpyAsync).  Inside a stream capture this becomes a memcpy node that reads `src`
// at every replay: the caller keeps the host array alive and unchanged for as long as the graph lives (ops._upload_table).
extern "C" int vcv_upload_table(void* dst, const void* src, int64_t bytes, void* stream) {
  if (!dst || !src || bytes <= 0) return VCV_EINVAL;
  return hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, (hipStream_t)stream) == hipSuccess ? VCV_OK : VCV_EHIP;
}

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
extern "C" int vcv_pack_many(VcvPackJob* jobs, int n, void* table_dev, void* stream) {
  if (!jobs || n <= 0 || !table_dev) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long long blocks = 0;
  for (int i = 0; i < n; ++i) {
    if (!jobs[i].w || !jobs[i].wp || jobs[i].total <= 0 || jobs[i].kind < 0 || jobs[i].kind > 2) return VCV_EINVAL;
    jobs[i].block0 = blocks;
    blocks += (jobs[i].total + 255) / 256;
  }
  if (blocks >= (1ll << 31)) return VCV_EINVAL;
  if (hipMemcpyAsync(table_dev, jobs, sizeof(VcvPackJob) * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(pack_many_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const VcvPackJob*)table_dev, n);
  return vcv_check_launch();
}

// stft.hip -- Hann-window STFT magnitude, forward and backward (n_fft = 2048).
//
// Replaces torch.stft / torchaudio.functional.spectrogram + sqrt(re^2+im^2+1e-6) at
// vits/mel_processing.py:54-96 of the reference (zero-pad variant :76-96 used in training,
// reflect-pad variant :54-74 / :115-142 used in validation).
//
// One workgroup (4 wavefronts) transforms NF consecutive frames of one utterance.  The 2048-point FFT is a radix-2
// decimation-in-frequency transform over 11 index bits held as  [wave : 2][lane : 6][register : 3]:
//   * stages of bits 10, 9, 8: every thread holds the 8 elements n = t + 256 r, so these butterflies are register-local;
//   * ONE exchange through LDS re-deals the elements so that bits 7..2 are the LANE index;
//   * stages of bits 7..2: wavefront shuffles (lane ^ 32, 16, 8, 4, 2, 1) -- the twiddle products are reduced across
//     the wavefront without touching LDS or a barrier (north_star: "wavefront shuffles for the STFT twiddle
//     reductions");
//   * stages of bits 1, 0: register-local again (twiddles 1 and -i).
// Two barriers per transform instead of the eleven of a Stockham pass through LDS.  The result is in bit-reversed
// order (element n holds X[rev11(n)]), which the consumers absorb in their index maps.  Twiddles come from a
// host-built fp64-accurate table staged in LDS.
// Frames are windowed straight from the un-padded waveform (the pad is folded into the index map -- no padded copy).
// Backward recomputes the frame spectrum (cheaper than saving re/im: the kernel is HBM-bound), forms
// G_k = dmag_k * X_k / mag_k, runs the conjugate-twiddle transform (the adjoint of the one-sided real DFT) and
// overlap-adds window * Re(g).  Zero-pad mode (the only one the training step differentiates): every workgroup OWNS the
// samples of its NF hops and gathers all frames that overlap them (3 halo frames per side are recomputed), so the
// gradient is written with plain stores -- no atomics, bit-reproducible.  Reflect mode keeps fp32 atomics.
#include "common.h"

namespace {

constexpr int N = 2048, HALF = 1024, NBIN = 1025, NF = 8, NT = 256;
constexpr int TWS = HALF + HALF / 32;  // twiddle table in LDS, skewed by one entry per 32 (see fft2048_ws)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// position of register r of this thread after the LDS re-deal:  n = wave*512 + (r>>2)*256 + lane*4 + (r&3)
__device__ __forceinline__ int pos2(int r, int t) {
  return ((t >> 6) << 9) | ((r >> 2) << 8) | ((t & 63) << 2) | (r & 3);
}
// frequency index held at position n after the transform
__device__ __forceinline__ int rev11(int n) { return (int)(__brev((unsigned)n) >> 21); }

// In: x[r] = element n = t + 256 r (natural order).  Out: x[r] = X[rev11(pos2(r, tid))].  `xch` = 2048 float2 of LDS scratch
// (the caller guarantees nobody still reads it), `tw` = LDS table exp(-2 pi i k / 2048), k < 1024.
// t = index of the thread within the 256 threads that share the frame (a workgroup may hold several such groups, each with
// its own `xch`; the barrier inside is the workgroup's, so all groups call this together)
template <bool INV>
__device__ __forceinline__ void fft2048_ws(float2 (&x)[8], float2* xch, const float2* tw, int t) {
  const int lane = t & 63;
  auto W = [&](int idx) {
    // (skewed table: the shuffle stages read twiddles 2^s entries apart -- 32 lanes on one bank of a dense table)
    float2 w = tw[idx + (idx >> 5)];
    if (INV) w.y = -w.y;
    return w;
  };
  // ---- bits 10, 9, 8: register-local ----
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // D = 1024: pairs (r, r+4), twiddle index n mod 1024 = t + 256 r
    const float2 a = x[r], b = x[r + 4];
    x[r] = cadd(a, b);
    x[r + 4] = cmul(csub(a, b), W(t + 256 * r));
  }
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 2; ++r) {  // D = 512: pairs (4g + r, 4g + r + 2), index (n mod 512) * 2
      const float2 a = x[4 * g + r], b = x[4 * g + r + 2];
      x[4 * g + r] = cadd(a, b);
      x[4 * g + r + 2] = cmul(csub(a, b), W((t + 256 * r) * 2));
    }
  {
    const float2 w = W(t * 4);  // D = 256: pairs (2g, 2g + 1), index (n mod 256) * 4
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float2 a = x[2 * g], b = x[2 * g + 1];
      x[2 * g] = cadd(a, b);
      x[2 * g + 1] = cmul(csub(a, b), w);
    }
  }
  // ---- re-deal through LDS: bits 7..2 become the lane ----
#pragma unroll
  for (int r = 0; r < 8; ++r) xch[t + 256 * r] = x[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) x[r] = xch[pos2(r, t)];
  // ---- bits 7..2: wavefront shuffles.  Stage of bit b = lane bit b - 2: the lane with that bit clear keeps a + b, its
  // partner (bit set) keeps (a - b) * W[(n mod 2^b) << (10 - b)] ----
#pragma unroll
  for (int lb = 5; lb >= 0; --lb) {
    const int m = 1 << lb, b = lb + 2;
    const bool upper = (lane & m) != 0;
    const int low = (lane & (m - 1)) << 2;  // bits of n below b that come from the lane
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float2 mine = x[r];
      const float2 other = make_float2(__shfl_xor(mine.x, m, 64), __shfl_xor(mine.y, m, 64));
      const float2 w = W((low | (r & 3)) << (10 - b));
      const float2 lo = cadd(mine, other);            // what the lower lane keeps (mine = a, other = b)
      const float2 hi = cmul(csub(other, mine), w);   // what the upper lane keeps (other = a, mine = b)
      x[r] = upper ? hi : lo;
    }
  }
  // ---- bits 1, 0: register-local.  D = 2: pairs (r, r+2) within a group of 4, twiddle 1 for even n, W[512] = -+i for odd ----
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    {
      const float2 a = x[4 * g], b = x[4 * g + 2];
      x[4 * g] = cadd(a, b);
      x[4 * g + 2] = csub(a, b);
    }
    {
      const float2 a = x[4 * g + 1], b = x[4 * g + 3];
      const float2 d = csub(a, b);
      x[4 * g + 1] = cadd(a, b);
      x[4 * g + 3] = INV ? make_float2(-d.y, d.x) : make_float2(d.y, -d.x);  // d * (+i) / d * (-i)
    }
#pragma unroll
    for (int r = 0; r < 4; r += 2) {  // D = 1
      const float2 a = x[4 * g + r], b = x[4 * g + r + 1];
      x[4 * g + r] = cadd(a, b);
      x[4 * g + r + 1] = csub(a, b);
    }
  }
}

// original-sample index of padded position pi (pad on both sides); -1 = zero
__device__ __forceinline__ int src_index(int pi, int pad, int T, int reflect) {
  int o = pi - pad;
  if (o >= 0 && o < T) return o;
  if (!reflect) return -1;
  if (o < 0) o = -o;
  else o = 2 * T - 2 - o;
  return (o >= 0 && o < T) ? o : -1;
}

// windowed frame starting at padded position `start`, in the transform's input layout (x[r] = element t + 256 r)
__device__ __forceinline__ void load_frame(float2 (&x)[8], const float* yb, const float* window, int start, int pad, int T,
                                           int reflect) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int n = threadIdx.x + 256 * r;
    const int o = src_index(start + n, pad, T, reflect);
    x[r] = make_float2(o >= 0 ? yb[o] * window[n] : 0.f, 0.f);
  }
}

// Forward.  A workgroup holds GROUPS groups of 256 threads; each group transforms FPG consecutive frames one after the other,
// so a workgroup covers NFW = GROUPS * FPG consecutive frames of one utterance and writes them as rows of NFW contiguous
// floats.  The window lives in registers (a thread always multiplies positions tid + 256 r), and the samples of the next
// frame are loaded before the current frame's transform starts, so their latency hides under it.  Round 3's kernel ran
// one group and eight frames per workgroup with a dependent global load in front of every transform: 64 workgroups for
// the training step's 16 x 32-frame launch, each a chain of eight load -> transform latencies (49 us, 75 GB/s).  Small
// launches now take one frame per group and workgroup (512 workgroups for that launch), large ones 4 groups x 4 frames.
template <int GROUPS, int FPG>
__global__ void __launch_bounds__(NT * GROUPS)
stft_mag_fwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                    const float2* __restrict__ twg, float* __restrict__ mag, int T, int F, int hop,
                    int pad, int reflect, float eps) {
  constexpr int NFW = GROUPS * FPG;
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  float2* tw = reinterpret_cast<float2*>(smem_);          // [HALF]
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
  float2* xch = tw + TWS + grp * N;                       // [GROUPS][N]
  float* tile = reinterpret_cast<float*>(tw + TWS + GROUPS * N);  // [NBIN][NFW]
  const int b = blockIdx.y, f0 = blockIdx.x * NFW;
  const float* yb = y + (size_t)b * T;
  for (int i = threadIdx.x; i < HALF; i += NT * GROUPS) tw[i + (i >> 5)] = twg[i];
  float win[8], raw[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) win[r] = window[tid + 256 * r];
  auto load_raw = [&](int f) {
    const int start = f * hop;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int o = f < F ? src_index(start + tid + 256 * r, pad, T, reflect) : -1;
      raw[r] = o >= 0 ? yb[o] : 0.f;
    }
  };
  load_raw(f0 + grp * FPG);
#pragma unroll
  for (int i = 0; i < FPG; ++i) {
    const int fi = grp * FPG + i;
    float2 x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = make_float2(raw[r] * win[r], 0.f);
    if (i + 1 < FPG) load_raw(f0 + fi + 1);  // in flight under this frame's transform
    __syncthreads();  // the previous frame's re-deal reads are done (and, first time round, tw is staged)
    fft2048_ws<false>(x, xch, tw, tid);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = rev11(pos2(r, tid));
      if (k < NBIN) tile[k * NFW + fi] = sqrtf(x[r].x * x[r].x + x[r].y * x[r].y + eps);
    }
  }
  __syncthreads();
  float* mb = mag + (size_t)b * NBIN * F;
  int nf = F - f0;
  if (nf > NFW) nf = NFW;
  if (NFW % 4 == 0 && (F & 3) == 0) {
    // rows of NFW floats, four frames per thread: 16-byte stores (f0 and F are multiples of four)
    for (int i = threadIdx.x; i < NBIN * (NFW / 4); i += NT * GROUPS) {
      const int k = i / (NFW / 4), q = i - k * (NFW / 4);
      if (4 * q < nf) *reinterpret_cast<f32x4*>(mb + (size_t)k * F + f0 + 4 * q) = *reinterpret_cast<const f32x4*>(tile + k * NFW + 4 * q);
    }
  } else {
    for (int i = threadIdx.x; i < NBIN * NFW; i += NT * GROUPS) {
      const int k = i / NFW, fi = i - k * NFW;
      if (fi < nf) mb[(size_t)k * F + f0 + fi] = tile[i];
    }
  }
}

// Backward, zero pad (the only mode the training step differentiates).  The workgroup OWNS the padded positions
// [f0 * hop, (f0 + NFO) * hop) (the last one also the tail) and gathers every frame that overlaps them (3 halo frames are
// recomputed); its GROUPS groups of 256 threads take those frames round robin, each accumulating into its OWN copy of the
// owned span in LDS (within a group the frames come one after the other: one thread per position and frame, no conflict);
// at the end the copies are added in group order and written with plain stores -- no atomics, no memset, bit-reproducible.
template <int GROUPS, int NFO>
__global__ void __launch_bounds__(NT * GROUPS)
stft_mag_bwd_own_kernel(const float* __restrict__ y, const float* __restrict__ window,
                        const float2* __restrict__ twg, const float* __restrict__ dmag,
                        float* __restrict__ dy, int T, int F, int hop, int pad, float eps) {
  constexpr int SPAN = NFO * 512 + N;  // hop <= 512 (checked by the launcher)
  extern __shared__ __attribute__((aligned(16))) char smem_[];
  float2* tw = reinterpret_cast<float2*>(smem_);
  const int grp = threadIdx.x >> 8, tid = threadIdx.x & 255;
  float2* xch = tw + TWS + grp * N;
  float* accs_all = reinterpret_cast<float*>(tw + TWS + GROUPS * N);  // [GROUPS][SPAN]
  float* accs = accs_all + grp * SPAN;
  const int b = blockIdx.y, f0 = blockIdx.x * NFO;
  const float* yb = y + (size_t)b * T;
  float* dyb = dy + (size_t)b * T;
  const float* db = dmag + (size_t)b * NBIN * F;
  for (int i = threadIdx.x; i < HALF; i += NT * GROUPS) tw[i + (i >> 5)] = twg[i];
  for (int i = threadIdx.x; i < GROUPS * SPAN; i += NT * GROUPS) accs_all[i] = 0.f;
  const bool last = (int)blockIdx.x == (int)gridDim.x - 1;
  const int own_lo = f0 * hop;
  const int own_hi = last ? T + 2 * pad : (f0 + NFO) * hop;  // padded positions [own_lo, own_hi)
  const int halo = (N + hop - 1) / hop - 1;  // frames starting up to N - 1 positions earlier still reach own_lo
  const int fa = f0 - halo < 0 ? 0 : f0 - halo;
  const int fb = f0 + NFO > F ? F : f0 + NFO;
  float win[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) win[r] = window[tid + 256 * r];
  for (int fr = fa; fr < fb; fr += GROUPS) {  // (every group runs every round: the transforms contain workgroup barriers)
    const int f = fr + grp;
    const bool live = f < fb;
    const int start = f * hop;
    float2 x[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int o = live ? src_index(start + tid + 256 * r, pad, T, 0) : -1;
      x[r] = make_float2(o >= 0 ? yb[o] * win[r] : 0.f, 0.f);
    }
    float dv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // the dmag column of this frame, loaded under the first transform
      const int k = rev11(pos2(r, tid));
      dv[r] = (live && k < NBIN) ? db[(size_t)k * F + f] : 0.f;
    }
    __syncthreads();
    fft2048_ws<false>(x, xch, tw, tid);  // spectrum, bit-reversed
    float2 G[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float m = sqrtf(x[r].x * x[r].x + x[r].y * x[r].y + eps);
      const float sc = dv[r] / m;
      G[r] = make_float2(sc * x[r].x, sc * x[r].y);
    }
    __syncthreads();  // every thread is past its re-deal reads of xch
#pragma unroll
    for (int r = 0; r < 8; ++r) xch[rev11(pos2(r, tid))] = G[r];  // natural order for the second transform
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = xch[tid + 256 * r];
    __syncthreads();
    fft2048_ws<true>(x, xch, tw, tid);  // g, bit-reversed
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) xch[rev11(pos2(r, tid))].x = x[r].x;  // natural order for the overlap-add
    __syncthreads();
    if (live) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int n = tid + 256 * r;
        const int pi = start + n;
        if (pi >= own_lo && pi < own_hi && pi - own_lo < SPAN) accs[pi - own_lo] += xch[n].x * win[r];
      }
    }
  }
  __syncthreads();
  for (int pi = own_lo + (int)threadIdx.x; pi < own_hi; pi += NT * GROUPS) {
    const int o = pi - pad;
    if (o >= 0 && o < T) {
      float v = 0.f;
      if (pi - own_lo < SPAN) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) v += accs_all[g * SPAN + pi - own_lo];  // fixed order
      }
      dyb[o] = v;
    }
  }
}

// Backward, general form (reflect pad: validation only; zero pad when the owning form above does not apply).  OWN = true:
// one group per workgroup, as above; OWN = false: each block scatters its NF frames with atomics.
template <bool OWN>
__global__ void __launch_bounds__(NT)
stft_mag_bwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                    const float2* __restrict__ twg, const float* __restrict__ dmag,
                    float* __restrict__ dy, int T, int F, int hop, int pad, int reflect, float eps) {
  __shared__ float2 xch[N];
  __shared__ float2 tw[TWS];
  __shared__ float accs[OWN ? NF * 512 + N : 1];  // hop <= 512 (checked by the launcher)
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f0 = blockIdx.x * NF;
  const float* yb = y + (size_t)b * T;
  float* dyb = dy + (size_t)b * T;
  const float* db = dmag + (size_t)b * NBIN * F;
  for (int i = tid; i < HALF; i += NT) tw[i + (i >> 5)] = twg[i];
  const bool last = (int)blockIdx.x == (int)gridDim.x - 1;
  const int own_lo = f0 * hop;
  const int own_hi = last ? T + 2 * pad : (f0 + NF) * hop;  // padded positions [own_lo, own_hi)
  int fa = f0, fb = f0 + NF;
  if (OWN) {
    for (int i = tid; i < NF * 512 + N; i += NT) accs[i] = 0.f;
    const int halo = (N + hop - 1) / hop - 1;  // frames starting up to N - 1 positions earlier still reach own_lo
    fa = f0 - halo < 0 ? 0 : f0 - halo;
  }
  if (fb > F) fb = F;
  for (int f = fa; f < fb; ++f) {
    const int start = f * hop;
    float2 x[8];
    load_frame(x, yb, window, start, pad, T, reflect);
    __syncthreads();
    fft2048_ws<false>(x, xch, tw, tid);  // spectrum, bit-reversed
    float2 G[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int k = rev11(pos2(r, tid));
      G[r] = make_float2(0.f, 0.f);
      if (k < NBIN) {
        const float m = sqrtf(x[r].x * x[r].x + x[r].y * x[r].y + eps);
        const float s = db[(size_t)k * F + f] / m;
        G[r] = make_float2(s * x[r].x, s * x[r].y);
      }
    }
    __syncthreads();  // every thread is past its re-deal reads of xch
#pragma unroll
    for (int r = 0; r < 8; ++r) xch[rev11(pos2(r, tid))] = G[r];  // natural order for the second transform
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) x[r] = xch[tid + 256 * r];
    __syncthreads();
    fft2048_ws<true>(x, xch, tw, tid);  // g, bit-reversed
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) xch[rev11(pos2(r, tid))].x = x[r].x;  // natural order for the coalesced overlap-add
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int n = tid + 256 * r;
      const float v = xch[n].x * window[n];
      if (OWN) {
        const int pi = start + n;
        if (pi >= own_lo && pi < own_hi) accs[pi - own_lo] += v;  // one thread per position and frame: no conflict
      } else {
        const int o = src_index(start + n, pad, T, reflect);
        if (o >= 0) unsafeAtomicAdd(dyb + o, v);
      }
    }
  }
  if (OWN) {
    __syncthreads();
    for (int pi = own_lo + tid; pi < own_hi; pi += NT) {
      const int o = pi - pad;
      if (o >= 0 && o < T) dyb[o] = pi - own_lo < NF * 512 + N ? accs[pi - own_lo] : 0.f;
    }
  }
}

// complex spectrogram (torchaudio Spectrogram(power=None) of vits/model/pipeline.py:24-26): out[b,k,f] = (re, im)
__global__ void __launch_bounds__(NT)
stft_complex_fwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                        const float2* __restrict__ twg, float2* __restrict__ out, int T, int F, int hop, int pad,
                        int reflect) {
  __shared__ float2 xch[N];
  __shared__ float2 tw[TWS];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f = blockIdx.x;
  const float* yb = y + (size_t)b * T;
  for (int i = tid; i < HALF; i += NT) tw[i + (i >> 5)] = twg[i];
  float2 x[8];
  load_frame(x, yb, window, f * hop, pad, T, reflect);
  __syncthreads();
  fft2048_ws<false>(x, xch, tw, tid);
  float2* ob = out + (size_t)b * NBIN * F;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int k = rev11(pos2(r, tid));
    if (k < NBIN) ob[(size_t)k * F + f] = x[r];
  }
}

// inverse STFT, stage 1 (torch.istft as used by torchaudio InverseSpectrogram, pipeline.py:28,66): per
// frame irfft (conjugate-symmetric extension, imaginary parts of DC / Nyquist ignored) x window, overlap-added
// into ola[b, f*hop + n] with fp32 atomics
__global__ void __launch_bounds__(NT)
istft_ola_kernel(const float2* __restrict__ spec, const float* __restrict__ window, const float2* __restrict__ twg,
                 float* __restrict__ ola, int F, int hop, int L) {
  __shared__ float2 xch[N];
  __shared__ float2 tw[TWS];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f = blockIdx.x;
  const float2* sb = spec + (size_t)b * NBIN * F;
  for (int i = tid; i < HALF; i += NT) tw[i + (i >> 5)] = twg[i];
  float2 x[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int k = tid + 256 * r;
    float2 v;
    if (k <= HALF) {
      v = sb[(size_t)k * F + f];
      if (k == 0 || k == HALF) v.y = 0.f;
    } else {
      v = sb[(size_t)(N - k) * F + f];
      v.y = -v.y;
    }
    x[r] = v;
  }
  __syncthreads();
  fft2048_ws<true>(x, xch, tw, tid);
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; ++r) xch[rev11(pos2(r, tid))].x = x[r].x;
  __syncthreads();
  float* ob = ola + (size_t)b * L;
  const float inv = 1.f / N;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int n = tid + 256 * r;
    unsafeAtomicAdd(ob + (size_t)f * hop + n, xch[n].x * inv * window[n]);
  }
}

// stage 2: divide by the window envelope sum_f w^2[i - f*hop] and trim `trim` samples on the left (center=True)
__global__ void istft_norm_kernel(const float* __restrict__ ola, const float* __restrict__ window,
                                  float* __restrict__ out, int F, int hop, int L, int trim, int Tout, size_t n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int t = (int)(idx % Tout);
  const size_t b = idx / Tout;
  const int i = t + trim;
  float env = 0.f;
  int f_hi = i / hop;
  if (f_hi > F - 1) f_hi = F - 1;
  for (int f = f_hi; f >= 0 && i - f * hop < N; --f) {
    const float w = window[i - f * hop];
    env += w * w;
  }
  out[idx] = env > 1e-11f ? ola[b * L + i] / env : 0.f;
}

}  // namespace

extern "C" int vcv_stft_complex_fwd(const float* y, const float* window, const float* twiddle, float* out, int B,
                                    int T, int n_fft, int hop, int pad, int reflect, void* stream) {
  if (!y || !window || !twiddle || !out || B <= 0 || T <= 0 || n_fft != N || hop <= 0 || pad < 0) return VCV_EINVAL;
  if (reflect && pad > T - 1) return VCV_EINVAL;
  const int F = (T + 2 * pad - n_fft) / hop + 1;
  if (F <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(stft_complex_fwd_kernel, dim3(F, B), dim3(NT), 0, (hipStream_t)stream, y, window,
                     (const float2*)twiddle, (float2*)out, T, F, hop, pad, reflect);
  return vcv_check_launch();
}

// spec: complex [B, 1025, F]; ola: workspace [B, n_fft + hop*(F-1)] (overwritten); out: [B, Tout] with
// Tout = hop*(F-1) when center (n_fft/2 trimmed on both sides), else the full overlap-add length
extern "C" int vcv_istft(const float* spec, const float* window, const float* twiddle, float* ola, float* out,
                         int B, int F, int n_fft, int hop, int center, void* stream) {
  if (!spec || !window || !twiddle || !ola || !out || B <= 0 || F <= 0 || n_fft != N || hop <= 0) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int L = n_fft + hop * (F - 1);
  const int trim = center ? n_fft / 2 : 0;
  const int Tout = center ? hop * (F - 1) : L;
  if (Tout <= 0) return VCV_EINVAL;
  if (hipMemsetAsync(ola, 0, sizeof(float) * (size_t)B * L, st) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(F, B), dim3(NT), 0, st, (const float2*)spec, window,
                     (const float2*)twiddle, ola, F, hop, L);
  const size_t n = (size_t)B * Tout;
  hipLaunchKernelGGL(istft_norm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ola, window, out, F, hop,
                     L, trim, Tout, n);
  return vcv_check_launch();
}

template <int GROUPS, int FPG>
static int launch_fwd(const float* y, const float* window, const float* twiddle, float* mag, int B, int T, int F, int hop, int pad,
                      int reflect, float eps, hipStream_t st) {
  constexpr int NFW = GROUPS * FPG;
  const size_t lds = sizeof(float2) * (TWS + (size_t)GROUPS * N) + sizeof(float) * (size_t)NBIN * NFW;
  auto kern = stft_mag_fwd_kernel<GROUPS, FPG>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VCV_EHIP;
  hipLaunchKernelGGL(kern, dim3(vcv_cdiv(F, NFW), B), dim3(NT * GROUPS), lds, st, y, window, (const float2*)twiddle, mag, T, F, hop,
                     pad, reflect, eps);
  return vcv_check_launch();
}

extern "C" int vcv_stft_mag_fwd(const float* y, const float* window, const float* twiddle, float* mag,
                                int B, int T, int n_fft, int hop, int pad, int reflect, float eps,
                                void* stream) {
  if (!y || !window || !twiddle || !mag || B <= 0 || T <= 0 || n_fft != N || hop <= 0 || pad < 0)
    return VCV_EINVAL;
  if (reflect && pad > T - 1) return VCV_EINVAL;
  const int F = (T + 2 * pad - n_fft) / hop + 1;
  if (F <= 0) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long long frames = (long long)B * F;
  // few frames: one per workgroup (the chip has 256 CUs); more: wider rows per store, the next frame's loads in flight
  if (frames <= 1024) return launch_fwd<1, 1>(y, window, twiddle, mag, B, T, F, hop, pad, reflect, eps, st);
  if (frames <= 4096) return launch_fwd<4, 1>(y, window, twiddle, mag, B, T, F, hop, pad, reflect, eps, st);
  return launch_fwd<4, 4>(y, window, twiddle, mag, B, T, F, hop, pad, reflect, eps, st);
}

template <int GROUPS, int NFO>
static int launch_bwd_own(const float* y, const float* window, const float* twiddle, const float* dmag, float* dy, int B, int T,
                          int F, int hop, int pad, float eps, hipStream_t st) {
  const size_t lds = sizeof(float2) * (TWS + (size_t)GROUPS * N) + sizeof(float) * (size_t)GROUPS * (NFO * 512 + N);
  auto kern = stft_mag_bwd_own_kernel<GROUPS, NFO>;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return VCV_EHIP;
  hipLaunchKernelGGL(kern, dim3(vcv_cdiv(F, NFO), B), dim3(NT * GROUPS), lds, st, y, window, (const float2*)twiddle, dmag, dy, T, F,
                     hop, pad, eps);
  return vcv_check_launch();
}

extern "C" int vcv_stft_mag_bwd(const float* y, const float* window, const float* twiddle,
                                const float* dmag, float* dy, int B, int T, int n_fft, int hop, int pad,
                                int reflect, float eps, void* stream) {
  if (!y || !window || !twiddle || !dmag || !dy || B <= 0 || T <= 0 || n_fft != N || hop <= 0 || pad < 0)
    return VCV_EINVAL;
  if (reflect && pad > T - 1) return VCV_EINVAL;
  const int F = (T + 2 * pad - n_fft) / hop + 1;
  if (F <= 0) return VCV_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // zero pad: every sample is owned by one workgroup -> plain stores, no memset, bit-reproducible.  Few frames (the training
  // step's 16 x 32-frame segment batch): two hops per workgroup, four frame groups; more: eight hops, two groups.
  auto tail_fits = [&](int nfo) { return T + 2 * pad - ((vcv_cdiv(F, nfo) - 1) * nfo) * hop <= nfo * 512 + N; };
  if (!reflect && hop <= 512) {
    if ((long long)B * F <= 2048 && tail_fits(2)) return launch_bwd_own<4, 2>(y, window, twiddle, dmag, dy, B, T, F, hop, pad, eps, st);
    if (tail_fits(8)) return launch_bwd_own<2, 8>(y, window, twiddle, dmag, dy, B, T, F, hop, pad, eps, st);
  }
  if (hipMemsetAsync(dy, 0, sizeof(float) * (size_t)B * T, st) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(stft_mag_bwd_kernel<false>, dim3(vcv_cdiv(F, NF), B), dim3(NT), 0, st, y,
                     window, (const float2*)twiddle, dmag, dy, T, F, hop, pad, reflect, eps);
  return vcv_check_launch();
}

// stft.hip -- Hann-window STFT magnitude, forward and backward (n_fft = 2048).
//
// Replaces torch.stft / torchaudio.functional.spectrogram + sqrt(re^2+im^2+1e-6) at
// vits/mel_processing.py:54-96 of the reference (zero-pad variant :76-96 used in training,
// reflect-pad variant :54-74 / :115-142 used in validation).
//
// One workgroup transforms NF consecutive frames of one utterance: a frame is windowed into LDS
// straight from the un-padded waveform (the pad is folded into the index map -- no padded copy),
// run through an 11-stage Stockham radix-2 FFT ping-ponging between two LDS buffers (twiddles
// from a host-built fp64-accurate table staged in LDS), and its 1025 magnitudes are parked in an
// LDS tile [bin][frame] so the [B, 1025, F] output is written in frame-contiguous runs.
// Backward recomputes the frame spectrum (cheaper than saving re/im: the kernel is HBM-bound),
// forms G_k = dmag_k * X_k / mag_k, runs the conjugate-twiddle FFT (the adjoint of the one-sided
// real DFT) and overlap-adds window * Re(g) into the waveform gradient with fp32 atomics.
#include "common.h"

namespace {

constexpr int N = 2048, HALF = 1024, NBIN = 1025, NF = 8, NT = 256;

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// result lands in B (11 stages: A->B->A->...->B)
template <bool INV>
__device__ __forceinline__ void fft2048(float2* A, float2* Bf, const float2* tw) {
  float2* in = A;
  float2* out = Bf;
  const int tid = threadIdx.x;
  for (int Ns = 1; Ns < N; Ns <<= 1) {
    const int tstep = HALF / Ns;
#pragma unroll
    for (int i = 0; i < HALF / NT; ++i) {
      const int j = tid + NT * i;
      const int k = j & (Ns - 1);
      float2 w = tw[k * tstep];
      if (INV) w.y = -w.y;
      const float2 v0 = in[j];
      const float2 v1 = cmul(in[j + HALF], w);
      const int j0 = ((j - k) << 1) + k;
      out[j0] = make_float2(v0.x + v1.x, v0.y + v1.y);
      out[j0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
    }
    __syncthreads();
    float2* t = in; in = out; out = t;
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

__global__ void __launch_bounds__(NT)
stft_mag_fwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                    const float2* __restrict__ twg, float* __restrict__ mag, int T, int F, int hop,
                    int pad, int reflect, float eps) {
  __shared__ float2 A[N];
  __shared__ float2 Bf[N];
  __shared__ float2 tw[HALF];
  __shared__ float tile[NBIN * NF];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f0 = blockIdx.x * NF;
  const float* yb = y + (size_t)b * T;
  for (int i = tid; i < HALF; i += NT) tw[i] = twg[i];
  int nf = F - f0;
  if (nf > NF) nf = NF;
  for (int fi = 0; fi < nf; ++fi) {
    const int start = (f0 + fi) * hop;
    __syncthreads();
    for (int n = tid; n < N; n += NT) {
      const int o = src_index(start + n, pad, T, reflect);
      A[n] = make_float2(o >= 0 ? yb[o] * window[n] : 0.f, 0.f);
    }
    __syncthreads();
    fft2048<false>(A, Bf, tw);
    for (int k = tid; k < NBIN; k += NT) {
      const float2 X = Bf[k];
      tile[k * NF + fi] = sqrtf(X.x * X.x + X.y * X.y + eps);
    }
  }
  __syncthreads();
  float* mb = mag + (size_t)b * NBIN * F;
  for (int i = tid; i < NBIN * NF; i += NT) {
    const int k = i / NF, fi = i - k * NF;
    if (fi < nf) mb[(size_t)k * F + f0 + fi] = tile[i];
  }
}

__global__ void __launch_bounds__(NT)
stft_mag_bwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                    const float2* __restrict__ twg, const float* __restrict__ dmag,
                    float* __restrict__ dy, int T, int F, int hop, int pad, int reflect, float eps) {
  __shared__ float2 A[N];
  __shared__ float2 Bf[N];
  __shared__ float2 tw[HALF];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f0 = blockIdx.x * NF;
  const float* yb = y + (size_t)b * T;
  float* dyb = dy + (size_t)b * T;
  const float* db = dmag + (size_t)b * NBIN * F;
  for (int i = tid; i < HALF; i += NT) tw[i] = twg[i];
  int nf = F - f0;
  if (nf > NF) nf = NF;
  for (int fi = 0; fi < nf; ++fi) {
    const int f = f0 + fi;
    const int start = f * hop;
    __syncthreads();
    for (int n = tid; n < N; n += NT) {
      const int o = src_index(start + n, pad, T, reflect);
      A[n] = make_float2(o >= 0 ? yb[o] * window[n] : 0.f, 0.f);
    }
    __syncthreads();
    fft2048<false>(A, Bf, tw);  // spectrum in Bf
    for (int k = tid; k < N; k += NT) {
      float2 G = make_float2(0.f, 0.f);
      if (k < NBIN) {
        const float2 X = Bf[k];
        const float m = sqrtf(X.x * X.x + X.y * X.y + eps);
        const float s = db[(size_t)k * F + f] / m;
        G = make_float2(s * X.x, s * X.y);
      }
      A[k] = G;
    }
    __syncthreads();
    fft2048<true>(A, Bf, tw);  // g in Bf
    for (int n = tid; n < N; n += NT) {
      const int o = src_index(start + n, pad, T, reflect);
      if (o >= 0) unsafeAtomicAdd(dyb + o, Bf[n].x * window[n]);
    }
  }
}

}  // namespace

extern "C" int vcv_stft_mag_fwd(const float* y, const float* window, const float* twiddle, float* mag,
                                int B, int T, int n_fft, int hop, int pad, int reflect, float eps,
                                void* stream) {
  if (!y || !window || !twiddle || !mag || B <= 0 || T <= 0 || n_fft != N || hop <= 0 || pad < 0)
    return VCV_EINVAL;
  if (reflect && pad > T - 1) return VCV_EINVAL;
  const int F = (T + 2 * pad - n_fft) / hop + 1;
  if (F <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(stft_mag_fwd_kernel, dim3(vcv_cdiv(F, NF), B), dim3(NT), 0, (hipStream_t)stream, y,
                     window, (const float2*)twiddle, mag, T, F, hop, pad, reflect, eps);
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
  if (hipMemsetAsync(dy, 0, sizeof(float) * (size_t)B * T, (hipStream_t)stream) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(stft_mag_bwd_kernel, dim3(vcv_cdiv(F, NF), B), dim3(NT), 0, (hipStream_t)stream, y,
                     window, (const float2*)twiddle, dmag, dy, T, F, hop, pad, reflect, eps);
  return vcv_check_launch();
}

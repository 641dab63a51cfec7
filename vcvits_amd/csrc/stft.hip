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

// complex spectrogram (torchaudio Spectrogram(power=None) of vits/model/pipeline.py:24-26): out[b,k,f] = (re, im)
__global__ void __launch_bounds__(NT)
stft_complex_fwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                        const float2* __restrict__ twg, float2* __restrict__ out, int T, int F, int hop, int pad,
                        int reflect) {
  __shared__ float2 A[N];
  __shared__ float2 Bf[N];
  __shared__ float2 tw[HALF];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f = blockIdx.x;
  const float* yb = y + (size_t)b * T;
  for (int i = tid; i < HALF; i += NT) tw[i] = twg[i];
  const int start = f * hop;
  for (int n = tid; n < N; n += NT) {
    const int o = src_index(start + n, pad, T, reflect);
    A[n] = make_float2(o >= 0 ? yb[o] * window[n] : 0.f, 0.f);
  }
  __syncthreads();
  fft2048<false>(A, Bf, tw);
  float2* ob = out + (size_t)b * NBIN * F;
  for (int k = tid; k < NBIN; k += NT) ob[(size_t)k * F + f] = Bf[k];
}

// inverse STFT, stage 1 (torch.istft as used by torchaudio InverseSpectrogram, pipeline.py:28,66): per
// frame irfft (conjugate-symmetric extension, imaginary parts of DC / Nyquist ignored) x window, overlap-added
// into ola[b, f*hop + n] with fp32 atomics
__global__ void __launch_bounds__(NT)
istft_ola_kernel(const float2* __restrict__ spec, const float* __restrict__ window, const float2* __restrict__ twg,
                 float* __restrict__ ola, int F, int hop, int L) {
  __shared__ float2 A[N];
  __shared__ float2 Bf[N];
  __shared__ float2 tw[HALF];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, f = blockIdx.x;
  const float2* sb = spec + (size_t)b * NBIN * F;
  for (int i = tid; i < HALF; i += NT) tw[i] = twg[i];
  for (int k = tid; k < N; k += NT) {
    float2 v;
    if (k <= HALF) {
      v = sb[(size_t)k * F + f];
      if (k == 0 || k == HALF) v.y = 0.f;
    } else {
      v = sb[(size_t)(N - k) * F + f];
      v.y = -v.y;
    }
    A[k] = v;
  }
  __syncthreads();
  fft2048<true>(A, Bf, tw);
  float* ob = ola + (size_t)b * L;
  const float inv = 1.f / N;
  for (int n = tid; n < N; n += NT) unsafeAtomicAdd(ob + (size_t)f * hop + n, Bf[n].x * inv * window[n]);
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

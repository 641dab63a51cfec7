// stft_generic.hip -- magnitude STFT forward / backward, complex STFT and inverse STFT for any power-of-two n_fft in [64, 4096]
// (radix-2 passes) and any other even n_fft in [16, 4096] (direct DFT, at the end of the file) other than the 2048 the two
// reference configs use (stft.hip holds the kernels tuned for that size).  Reference: vits/mel_processing.py:54-96
// (spectrogram_torch / spectrogram_torch_audio take n_fft, hop_size, win_size as arguments; torch.stft pads a shorter window
// to n_fft, centred -- the `window` table handed in is already that padded window).
//
// One workgroup of 256 threads per FR consecutive frames of one row: a frame's windowed samples go into LDS in bit-reversed
// order, log2(n_fft) radix-2 passes with the [n_fft/2] twiddle table (cos, -sin), magnitudes of the FR frames are staged in
// LDS and written as FR consecutive floats per bin.  Backward: the frame's transform again, G_k = dmag_k X_k / mag_k on the
// one-sided bins, the same transform on conj(G) (whose real part is the un-normalised inverse), times the window, added into
// dy with fp32 atomics (frames overlap; dy is zeroed by the launcher).  A cold path: correct and coalesced, not tuned.
#include "common.h"

namespace {

constexpr int GT = 256;  // threads per workgroup
constexpr int FR = 4;    // frames per workgroup (forward)

__device__ __forceinline__ int sample_index(int o, int T, int reflect) {
  if (o >= 0 && o < T) return o;
  if (!reflect) return -1;
  return o < 0 ? -o : 2 * (T - 1) - o;  // (pad <= T - 1 checked by the launcher)
}

// in-place radix-2 decimation-in-time passes over buf[n_fft] (input in bit-reversed order, output in natural order)
__device__ __forceinline__ void fft_passes(float2* buf, const float2* __restrict__ tw, int n_fft, int logn) {
  for (int s = 1; s <= logn; ++s) {
    const int half = 1 << (s - 1);
    const int tstep = n_fft >> s;
    __syncthreads();
    for (int j = threadIdx.x; j < n_fft / 2; j += GT) {
      const int k = j & (half - 1);
      const int i0 = ((j >> (s - 1)) << s) + k;
      const int i1 = i0 + half;
      const float2 w = tw[k * tstep];
      const float2 a = buf[i0], b = buf[i1];
      const float2 t = make_float2(w.x * b.x - w.y * b.y, w.x * b.y + w.y * b.x);
      buf[i0] = make_float2(a.x + t.x, a.y + t.y);
      buf[i1] = make_float2(a.x - t.x, a.y - t.y);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void load_frame(float2* buf, const float* __restrict__ yb, const float* __restrict__ window, int f,
                                           int hop, int pad, int T, int reflect, int n_fft, int logn) {
  for (int n = threadIdx.x; n < n_fft; n += GT) {
    const int o = sample_index(f * hop + n - pad, T, reflect);
    const float v = o >= 0 ? yb[o] * window[n] : 0.f;
    buf[__brev((unsigned)n) >> (32 - logn)] = make_float2(v, 0.f);
  }
}

__global__ void __launch_bounds__(GT) stft_mag_fwd_generic_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                                 const float2* __restrict__ tw, float* __restrict__ mag, int T,
                                                                 int F, int hop, int pad, int reflect, float eps, int n_fft,
                                                                 int logn) {
  extern __shared__ float2 smem[];
  float2* buf = smem;                                  // [n_fft]
  float* outs = reinterpret_cast<float*>(smem + n_fft);  // [n_fft/2 + 1][FR]
  const int nbin = n_fft / 2 + 1;
  const int b = blockIdx.y, f0 = blockIdx.x * FR;
  const float* yb = y + (size_t)b * T;
  const int nf = min(FR, F - f0);
  for (int i = 0; i < nf; ++i) {
    __syncthreads();
    load_frame(buf, yb, window, f0 + i, hop, pad, T, reflect, n_fft, logn);
    fft_passes(buf, tw, n_fft, logn);
    for (int k = threadIdx.x; k < nbin; k += GT) {
      const float2 x = buf[k];
      outs[k * FR + i] = sqrtf(x.x * x.x + x.y * x.y + eps);
    }
  }
  __syncthreads();
  float* mb = mag + (size_t)b * nbin * F + f0;
  for (int e = threadIdx.x; e < nbin * FR; e += GT) {
    const int k = e / FR, i = e % FR;
    if (i < nf) mb[(size_t)k * F + i] = outs[e];
  }
}

__global__ void __launch_bounds__(GT) stft_mag_bwd_generic_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                                 const float2* __restrict__ tw, const float* __restrict__ dmag,
                                                                 float* __restrict__ dy, int T, int F, int hop, int pad,
                                                                 int reflect, float eps, int n_fft, int logn) {
  extern __shared__ float2 smem[];
  float2* buf = smem;
  const int nbin = n_fft / 2 + 1;
  const int b = blockIdx.y, f = blockIdx.x;
  const float* yb = y + (size_t)b * T;
  load_frame(buf, yb, window, f, hop, pad, T, reflect, n_fft, logn);
  fft_passes(buf, tw, n_fft, logn);
  // G_k = dmag_k * X_k / mag_k for the one-sided bins (<= 4096 / 2 / 256 + 1 = 9 per thread), then conj(G) back into the
  // buffer in bit-reversed order, zeros above the Nyquist bin
  float2 g[9];
  const float* db = dmag + (size_t)b * nbin * F + f;
  for (int r = 0; r < 9; ++r) {
    const int k = threadIdx.x + GT * r;
    g[r] = make_float2(0.f, 0.f);
    if (k < nbin) {
      const float2 x = buf[k];
      const float sc = db[(size_t)k * F] / sqrtf(x.x * x.x + x.y * x.y + eps);
      g[r] = make_float2(x.x * sc, -x.y * sc);
    }
  }
  __syncthreads();
  for (int n = threadIdx.x; n < n_fft; n += GT) buf[n] = make_float2(0.f, 0.f);
  __syncthreads();
  for (int r = 0; r < 9; ++r) {
    const int k = threadIdx.x + GT * r;
    if (k < nbin) buf[__brev((unsigned)k) >> (32 - logn)] = g[r];
  }
  fft_passes(buf, tw, n_fft, logn);
  float* dyb = dy + (size_t)b * T;
  for (int n = threadIdx.x; n < n_fft; n += GT) {
    const int o = sample_index(f * hop + n - pad, T, reflect);
    if (o >= 0) unsafeAtomicAdd(dyb + o, buf[n].x * window[n]);
  }
}

// complex spectrum of one frame per workgroup: out [B, n_fft/2+1, F] complex (the source-audio pipeline's Spectrogram)
__global__ void __launch_bounds__(GT) stft_complex_fwd_generic_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                                     const float2* __restrict__ tw, float2* __restrict__ out,
                                                                     int T, int F, int hop, int pad, int reflect, int n_fft,
                                                                     int logn) {
  extern __shared__ float2 smem[];
  float2* buf = smem;
  const int nbin = n_fft / 2 + 1;
  const int b = blockIdx.y, f = blockIdx.x;
  load_frame(buf, y + (size_t)b * T, window, f, hop, pad, T, reflect, n_fft, logn);
  fft_passes(buf, tw, n_fft, logn);
  float2* ob = out + (size_t)b * nbin * F + f;
  for (int k = threadIdx.x; k < nbin; k += GT) ob[(size_t)k * F] = buf[k];
}

// inverse STFT, stage 1 (torch.istft): conjugate-symmetric extension of the one-sided frame (imaginary parts of DC / Nyquist
// ignored), inverse transform = Re FFT(conj V) / n_fft, times the window, overlap-added into ola with fp32 atomics
__global__ void __launch_bounds__(GT) istft_ola_generic_kernel(const float2* __restrict__ spec, const float* __restrict__ window,
                                                              const float2* __restrict__ tw, float* __restrict__ ola, int F,
                                                              int hop, int L, int n_fft, int logn) {
  extern __shared__ float2 smem[];
  float2* buf = smem;
  const int nbin = n_fft / 2 + 1, half = n_fft / 2;
  const int b = blockIdx.y, f = blockIdx.x;
  const float2* sb = spec + (size_t)b * nbin * F + f;
  for (int k = threadIdx.x; k < n_fft; k += GT) {
    float2 v;
    if (k <= half) {
      v = sb[(size_t)k * F];
      v.y = (k == 0 || k == half) ? 0.f : -v.y;  // conj(V_k)
    } else {
      v = sb[(size_t)(n_fft - k) * F];           // V_k = conj(V_{n-k}), so conj(V_k) = V_{n-k}
    }
    buf[__brev((unsigned)k) >> (32 - logn)] = v;
  }
  fft_passes(buf, tw, n_fft, logn);
  float* ob = ola + (size_t)b * L + (size_t)f * hop;
  const float inv = 1.f / (float)n_fft;
  for (int n = threadIdx.x; n < n_fft; n += GT) unsafeAtomicAdd(ob + n, buf[n].x * inv * window[n]);
}

// ---- any other EVEN n_fft (e.g. 1280): direct DFT, O(n_fft) per bin.  x[n] (windowed frame) and the [n_fft/2] twiddle table
// live in LDS; e^{-2 pi i q / n} for q >= n/2 is minus the entry q - n/2; q = k n mod n_fft is kept by addition.  Sums are
// carried in double (a direct sum of up to 4096 terms in fp32 would miss the 1e-5 bound the transforms are held to).
struct Dft {
  float* xs;     // [n_fft]
  float2* tws;   // [n_fft / 2]
  float2* spec;  // [n_fft / 2 + 1]
};

__device__ __forceinline__ Dft dft_lds(float2* smem, int n_fft) {
  Dft d;
  d.tws = smem;
  d.spec = smem + n_fft / 2;
  d.xs = reinterpret_cast<float*>(smem + n_fft / 2 + n_fft / 2 + 1);
  return d;
}

__device__ __forceinline__ float2 tw_at(const float2* tws, int q, int half) {
  const float2 w = tws[q >= half ? q - half : q];
  return q >= half ? make_float2(-w.x, -w.y) : w;
}

// xs <- windowed frame, tws <- table; then spec[k] = X_k for the one-sided bins
__device__ __forceinline__ void dft_frame(const Dft& d, const float* __restrict__ yb, const float* __restrict__ window,
                                          const float2* __restrict__ tw, int f, int hop, int pad, int T, int reflect, int n_fft) {
  const int half = n_fft / 2;
  for (int n = threadIdx.x; n < n_fft; n += GT) {
    const int o = sample_index(f * hop + n - pad, T, reflect);
    d.xs[n] = o >= 0 ? yb[o] * window[n] : 0.f;
  }
  for (int i = threadIdx.x; i < half; i += GT) d.tws[i] = tw[i];
  __syncthreads();
  for (int k = threadIdx.x; k <= half; k += GT) {
    double re = 0.0, im = 0.0;
    int q = 0;
    for (int n = 0; n < n_fft; ++n) {
      const float2 w = tw_at(d.tws, q, half);
      const float x = d.xs[n];
      re += (double)(x * w.x);
      im += (double)(x * w.y);
      q += k;
      if (q >= n_fft) q -= n_fft;
    }
    d.spec[k] = make_float2((float)re, (float)im);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(GT) stft_dft_fwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                         const float2* __restrict__ tw, float* __restrict__ mag,
                                                         float2* __restrict__ cplx, int T, int F, int hop, int pad, int reflect,
                                                         float eps, int n_fft) {
  extern __shared__ float2 smem[];
  const Dft d = dft_lds(smem, n_fft);
  const int nbin = n_fft / 2 + 1;
  const int b = blockIdx.y, f = blockIdx.x;
  dft_frame(d, y + (size_t)b * T, window, tw, f, hop, pad, T, reflect, n_fft);
  for (int k = threadIdx.x; k < nbin; k += GT) {
    const float2 x = d.spec[k];
    const size_t o = ((size_t)b * nbin + k) * F + f;
    if (mag) mag[o] = sqrtf(x.x * x.x + x.y * x.y + eps);
    else cplx[o] = x;
  }
}

__global__ void __launch_bounds__(GT) stft_dft_bwd_kernel(const float* __restrict__ y, const float* __restrict__ window,
                                                         const float2* __restrict__ tw, const float* __restrict__ dmag,
                                                         float* __restrict__ dy, int T, int F, int hop, int pad, int reflect,
                                                         float eps, int n_fft) {
  extern __shared__ float2 smem[];
  const Dft d = dft_lds(smem, n_fft);
  const int nbin = n_fft / 2 + 1, half = n_fft / 2;
  const int b = blockIdx.y, f = blockIdx.x;
  dft_frame(d, y + (size_t)b * T, window, tw, f, hop, pad, T, reflect, n_fft);
  for (int k = threadIdx.x; k < nbin; k += GT) {  // G_k = dmag_k X_k / mag_k
    const float2 x = d.spec[k];
    const float sc = dmag[((size_t)b * nbin + k) * F + f] / sqrtf(x.x * x.x + x.y * x.y + eps);
    d.spec[k] = make_float2(x.x * sc, x.y * sc);
  }
  __syncthreads();
  float* dyb = dy + (size_t)b * T;
  for (int n = threadIdx.x; n < n_fft; n += GT) {  // d/dx[n] = sum_k a_k cos(th) - b_k sin(th) = a_k w.x + b_k w.y
    const int o = sample_index(f * hop + n - pad, T, reflect);
    if (o < 0) continue;
    double acc = 0.0;
    int q = 0;
    for (int k = 0; k <= half; ++k) {
      const float2 w = tw_at(d.tws, q, half);
      const float2 g = d.spec[k];
      acc += (double)(g.x * w.x + g.y * w.y);
      q += n;
      if (q >= n_fft) q -= n_fft;
    }
    unsafeAtomicAdd(dyb + o, (float)acc * window[n]);
  }
}

__global__ void __launch_bounds__(GT) istft_dft_ola_kernel(const float2* __restrict__ spec, const float* __restrict__ window,
                                                          const float2* __restrict__ tw, float* __restrict__ ola, int F, int hop,
                                                          int L, int n_fft) {
  extern __shared__ float2 smem[];
  const Dft d = dft_lds(smem, n_fft);
  const int nbin = n_fft / 2 + 1, half = n_fft / 2;
  const int b = blockIdx.y, f = blockIdx.x;
  for (int k = threadIdx.x; k < nbin; k += GT) d.spec[k] = spec[((size_t)b * nbin + k) * F + f];
  for (int i = threadIdx.x; i < half; i += GT) d.tws[i] = tw[i];
  __syncthreads();
  float* ob = ola + (size_t)b * L + (size_t)f * hop;
  const double inv = 1.0 / (double)n_fft;
  for (int n = threadIdx.x; n < n_fft; n += GT) {
    // x[n] = (V_0.re + (-1)^n V_half.re + 2 sum_{0<k<half} (re_k cos(th) - im_k sin(th))) / n_fft, e^{+i th}: sin = -w.y
    double acc = (double)d.spec[0].x + ((n & 1) ? -1.0 : 1.0) * (double)d.spec[half].x;
    int q = n >= n_fft ? n - n_fft : n;
    for (int k = 1; k < half; ++k) {
      const float2 w = tw_at(d.tws, q, half);
      const float2 v = d.spec[k];
      acc += 2.0 * (double)(v.x * w.x + v.y * w.y);
      q += n;
      if (q >= n_fft) q -= n_fft;
    }
    unsafeAtomicAdd(ob + n, (float)(acc * inv) * window[n]);
  }
}

static inline bool dft_size(int n_fft) { return n_fft >= 16 && n_fft <= 4096 && (n_fft & 1) == 0; }
static inline size_t dft_lds_bytes(int n_fft) { return sizeof(float2) * (size_t)(n_fft + 1) + sizeof(float) * (size_t)n_fft; }

}  // namespace

int stft_complex_fwd_generic_launch(const float* y, const float* window, const float* twiddle, float* out, int B, int T, int n_fft,
                                    int hop, int pad, int reflect, hipStream_t st) {
  const int F = (T + 2 * pad - n_fft) / hop + 1;
  if (F <= 0) return VCV_EINVAL;
  if (n_fft < 64 || (n_fft & (n_fft - 1))) {
    if (!dft_size(n_fft)) return VCV_EINVAL;
    hipLaunchKernelGGL(stft_dft_fwd_kernel, dim3(F, B), dim3(GT), dft_lds_bytes(n_fft), st, y, window, (const float2*)twiddle,
                       (float*)nullptr, (float2*)out, T, F, hop, pad, reflect, 0.f, n_fft);
    return vcv_check_launch();
  }
  if (n_fft > 4096) return VCV_EINVAL;
  const int logn = 31 - __builtin_clz((unsigned)n_fft);
  hipLaunchKernelGGL(stft_complex_fwd_generic_kernel, dim3(F, B), dim3(GT), sizeof(float2) * (size_t)n_fft, st, y, window,
                     (const float2*)twiddle, (float2*)out, T, F, hop, pad, reflect, n_fft, logn);
  return vcv_check_launch();
}

// stage 1 of vcv_istft for n_fft != 2048 (ola zeroed by the caller; stage 2 = stft.hip's istft_norm_kernel)
int istft_ola_generic_launch(const float* spec, const float* window, const float* twiddle, float* ola, int B, int F, int n_fft,
                             int hop, int L, hipStream_t st) {
  if (n_fft < 64 || (n_fft & (n_fft - 1))) {
    if (!dft_size(n_fft)) return VCV_EINVAL;
    hipLaunchKernelGGL(istft_dft_ola_kernel, dim3(F, B), dim3(GT), dft_lds_bytes(n_fft), st, (const float2*)spec, window,
                       (const float2*)twiddle, ola, F, hop, L, n_fft);
    return vcv_check_launch();
  }
  if (n_fft > 4096) return VCV_EINVAL;
  const int logn = 31 - __builtin_clz((unsigned)n_fft);
  hipLaunchKernelGGL(istft_ola_generic_kernel, dim3(F, B), dim3(GT), sizeof(float2) * (size_t)n_fft, st, (const float2*)spec, window,
                     (const float2*)twiddle, ola, F, hop, L, n_fft, logn);
  return vcv_check_launch();
}

// called by vcv_stft_mag_fwd / vcv_stft_mag_bwd (stft.hip) for n_fft != 2048
int stft_mag_fwd_generic_launch(const float* y, const float* window, const float* twiddle, float* mag, int B, int T, int n_fft, int hop,
                             int pad, int reflect, float eps, hipStream_t st) {
  const int F = (T + 2 * pad - n_fft) / hop + 1;
  if (F <= 0) return VCV_EINVAL;
  if (n_fft < 64 || (n_fft & (n_fft - 1))) {
    if (!dft_size(n_fft)) return VCV_EINVAL;
    hipLaunchKernelGGL(stft_dft_fwd_kernel, dim3(F, B), dim3(GT), dft_lds_bytes(n_fft), st, y, window, (const float2*)twiddle, mag,
                       (float2*)nullptr, T, F, hop, pad, reflect, eps, n_fft);
    return vcv_check_launch();
  }
  if (n_fft > 4096) return VCV_EINVAL;
  const int logn = 31 - __builtin_clz((unsigned)n_fft);
  const size_t lds = sizeof(float2) * (size_t)n_fft + sizeof(float) * (size_t)(n_fft / 2 + 1) * FR;
  if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)stft_mag_fwd_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)lds) != hipSuccess)
    return VCV_EHIP;
  hipLaunchKernelGGL(stft_mag_fwd_generic_kernel, dim3(vcv_cdiv(F, FR), B), dim3(GT), lds, st, y, window, (const float2*)twiddle, mag,
                     T, F, hop, pad, reflect, eps, n_fft, logn);
  return vcv_check_launch();
}

int stft_mag_bwd_generic_launch(const float* y, const float* window, const float* twiddle, const float* dmag, float* dy, int B, int T,
                             int n_fft, int hop, int pad, int reflect, float eps, hipStream_t st) {
  const int F = (T + 2 * pad - n_fft) / hop + 1;
  if (F <= 0) return VCV_EINVAL;
  const bool dft = n_fft < 64 || (n_fft & (n_fft - 1));
  if (dft ? !dft_size(n_fft) : n_fft > 4096) return VCV_EINVAL;
  if (vcv_zero_async(dy, sizeof(float) * (size_t)B * T, st) != hipSuccess) return VCV_EHIP;
  if (dft) {
    hipLaunchKernelGGL(stft_dft_bwd_kernel, dim3(F, B), dim3(GT), dft_lds_bytes(n_fft), st, y, window, (const float2*)twiddle, dmag,
                       dy, T, F, hop, pad, reflect, eps, n_fft);
    return vcv_check_launch();
  }
  const int logn = 31 - __builtin_clz((unsigned)n_fft);
  hipLaunchKernelGGL(stft_mag_bwd_generic_kernel, dim3(F, B), dim3(GT), sizeof(float2) * (size_t)n_fft, st, y, window,
                     (const float2*)twiddle, dmag, dy, T, F, hop, pad, reflect, eps, n_fft, logn);
  return vcv_check_launch();
}

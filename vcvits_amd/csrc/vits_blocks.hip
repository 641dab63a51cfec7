// vits_blocks.hip -- the HBM-bound glue of the VITS blocks, one streaming pass each:
// WaveNet gate / residual-skip update (modules.py:147-175, commons.py:99-106), posterior
// sampling (posterior_encoder.py:36-38), mean-only coupling (modules.py:317-336), channel
// LayerNorm (modules.py:19-31), the banded relative-position softmax of the content encoder
// (relative_attention_transformer.py:157-180), KL loss (losses.py:40-55), nearest interpolation
// and segment slicing (synthesizer_svc.py:83-86, commons.py:48-64).
// The dense contractions around them run on the MFMA conv/GEMM kernels (conv_gemm.hip).
#include "common.h"

namespace {

__device__ __forceinline__ float wsum(float s) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  return s;
}
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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// counter-based dropout mask: scale 1/(1-p) with probability 1-p, else 0 (stateless, so forward,
// backward and the transposed copy of a tensor all regenerate the same mask from (seed, index))
__device__ __forceinline__ float drop_scale(unsigned long long seed, unsigned long long idx, float p, float inv_keep) {
  unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  return u >= p ? inv_keep : 0.f;
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float p,
                               float inv_keep, unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  if (seed_off) seed += *seed_off;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = x[i] * drop_scale(seed, i, p, inv_keep);
}

// ---- WN gate ---------------------------------------------------------------------------------
// acts[b,c,t] = tanh(xin[b,c,t] + g[b,goff+c]) * sigmoid(xin[b,H+c,t] + g[b,goff+H+c])
__global__ void wn_gate_fwd_kernel(const float* __restrict__ xin, const float* __restrict__ g, int gstride,
                                   int goff, float* __restrict__ acts, int H, int T, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t bc = i / T;
  const int c = (int)(bc % H);
  const size_t b = bc / H;
  const size_t ia = (b * 2 * H + c) * T + t, ib = ia + (size_t)H * T;
  float ga = 0.f, gb = 0.f;
  if (g) { ga = g[b * gstride + goff + c]; gb = g[b * gstride + goff + H + c]; }
  acts[i] = tanhf(xin[ia] + ga) * sigmoidf_(xin[ib] + gb);
}

__global__ void wn_gate_bwd_kernel(const float* __restrict__ xin, const float* __restrict__ g, int gstride,
                                   int goff, const float* __restrict__ dacts, float* __restrict__ dxin,
                                   int H, int T, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t bc = i / T;
  const int c = (int)(bc % H);
  const size_t b = bc / H;
  const size_t ia = (b * 2 * H + c) * T + t, ib = ia + (size_t)H * T;
  float ga = 0.f, gb = 0.f;
  if (g) { ga = g[b * gstride + goff + c]; gb = g[b * gstride + goff + H + c]; }
  const float th = tanhf(xin[ia] + ga), sg = sigmoidf_(xin[ib] + gb);
  const float d = dacts[i];
  dxin[ia] = d * sg * (1.f - th * th);
  dxin[ib] = d * th * sg * (1.f - sg);
}

// out[r*ostride + ooff... ] row sums: x [R, T] -> out[(r / inner) * ostride + ooff + r % inner]
__global__ void __launch_bounds__(64) row_sum_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                     int T, int inner, int ostride, int ooff) {
  const int r = blockIdx.x;
  const float* xr = x + (size_t)r * T;
  float s = 0.f;
  for (int t = threadIdx.x; t < T; t += 64) s += xr[t];
  s = wsum(s);
  if (threadIdx.x == 0) out[(size_t)(r / inner) * ostride + ooff + r % inner] = s;
}

// ---- WN residual / skip update ------------------------------------------------------------------
// last == 0: rs [B,2H,T]: x_new = (x + rs[:, :H]) * mask ; out_new = out + rs[:, H:]
// last == 1: rs [B,H,T]:  out_new = out + rs              (x untouched)
__global__ void wn_res_skip_fwd_kernel(const float* __restrict__ x, const float* __restrict__ out,
                                       const float* __restrict__ rs, const float* __restrict__ mask,
                                       float* __restrict__ xn, float* __restrict__ on, int H, int T, int last,
                                       size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t bc = i / T;
  const int c = (int)(bc % H);
  const size_t b = bc / H;
  const float o = out ? out[i] : 0.f;
  if (last) { on[i] = o + rs[i]; return; }
  const size_t ir = (b * 2 * H + c) * T + t;
  xn[i] = (x[i] + rs[ir]) * mask[b * T + t];
  on[i] = o + rs[ir + (size_t)H * T];
}

// drs from (dxn, don); dx = dxn * mask
__global__ void wn_res_skip_bwd_kernel(const float* __restrict__ dxn, const float* __restrict__ don,
                                       const float* __restrict__ mask, float* __restrict__ drs,
                                       float* __restrict__ dx, int H, int T, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t bc = i / T;
  const int c = (int)(bc % H);
  const size_t b = bc / H;
  const size_t ir = (b * 2 * H + c) * T + t;
  const float d = (dxn ? dxn[i] : 0.f) * mask[b * T + t];
  dx[i] = d;
  drs[ir] = d;
  drs[ir + (size_t)H * T] = don ? don[i] : 0.f;
}

// ---- stats split / posterior sampling -----------------------------------------------------------
// stats [B,2C,T] (un-masked conv output): m = stats[:, :C]*mask, logs = stats[:, C:]*mask,
// z = (m + eps*exp(logs))*mask (when eps != NULL)
__global__ void split_sample_fwd_kernel(const float* __restrict__ stats, const float* __restrict__ eps,
                                        const float* __restrict__ mask, float* __restrict__ m,
                                        float* __restrict__ logs, float* __restrict__ z, int C, int T, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t bc = i / T;
  const int c = (int)(bc % C);
  const size_t b = bc / C;
  const float mk = mask[b * T + t];
  const size_t im = (b * 2 * C + c) * T + t;
  const float mm = stats[im] * mk, ll = stats[im + (size_t)C * T] * mk;
  m[i] = mm; logs[i] = ll;
  if (eps) z[i] = (mm + eps[i] * expf(ll)) * mk;
}

__global__ void split_sample_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ dlogs,
                                        const float* __restrict__ dz, const float* __restrict__ eps,
                                        const float* __restrict__ logs, const float* __restrict__ mask,
                                        float* __restrict__ dstats, int C, int T, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t bc = i / T;
  const int c = (int)(bc % C);
  const size_t b = bc / C;
  const float mk = mask[b * T + t];
  float gm = dm ? dm[i] : 0.f, gl = dlogs ? dlogs[i] : 0.f;
  if (dz && eps) {
    const float gz = dz[i] * mk;
    gm += gz;
    gl += gz * eps[i] * expf(logs[i]);
  }
  const size_t im = (b * 2 * C + c) * T + t;
  dstats[im] = gm * mk;
  dstats[im + (size_t)C * T] = gl * mk;
}

// ---- mean-only coupling: fwd x1n = m + x1*mask ; reverse x1n = (x1 - m)*mask ---------------------
__global__ void coupling_kernel(const float* __restrict__ x1, const float* __restrict__ m,
                                const float* __restrict__ mask, float* __restrict__ y, int C, int T, int reverse,
                                size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t b = i / ((size_t)C * T);
  const float mk = mask[b * T + t];
  y[i] = reverse ? (x1[i] - m[i]) * mk : m[i] + x1[i] * mk;
}

// ---- prior sample of SynthesizerSVC.infer: z_p = m + noise * exp(logs) * noise_scale ------------------
__global__ void prior_sample_kernel(const float* __restrict__ m, const float* __restrict__ logs,
                                    const float* __restrict__ noise, float* __restrict__ z, float noise_scale, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  z[i] = m[i] + noise[i] * expf(logs[i]) * noise_scale;
}

// ---- channel LayerNorm of (x + y) over C for [B,C,T] -----------------------------------------------
// block: 64 consecutive t (lanes) x 4 channel groups (waves); two-pass mean / variance in registers.
__global__ void __launch_bounds__(256)
layernorm_c_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                       const float* __restrict__ beta, float* __restrict__ out, float* __restrict__ mean,
                       float* __restrict__ rstd, int C, int T, float eps) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
  const bool ok = t < T;
  const size_t base = (size_t)b * C * T + t;
  float s = 0.f;
  for (int c = wv; c < C; c += 4)
    if (ok) s += x[base + (size_t)c * T] + (y ? y[base + (size_t)c * T] : 0.f);
  red[wv][lane] = s;
  __syncthreads();
  const float mu = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / C;
  __syncthreads();
  float v = 0.f;
  for (int c = wv; c < C; c += 4)
    if (ok) { const float d = x[base + (size_t)c * T] + (y ? y[base + (size_t)c * T] : 0.f) - mu; v += d * d; }
  red[wv][lane] = v;
  __syncthreads();
  const float var = (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]) / C;
  const float rs = rsqrtf(var + eps);
  for (int c = wv; c < C; c += 4)
    if (ok) {
      const float d = x[base + (size_t)c * T] + (y ? y[base + (size_t)c * T] : 0.f) - mu;
      out[base + (size_t)c * T] = d * rs * gamma[c] + beta[c];
    }
  if (wv == 0 && ok) { mean[(size_t)b * T + t] = mu; rstd[(size_t)b * T + t] = rs; }
}

// C = 4 * CPW (32 / 64 channels per wave): x + y is loaded ONCE into registers (rounds of 8 channels x 2 tensors in flight),
// mean, variance and the output come from there -- the generic kernel above walks the channels three times, and its last
// pass (a store per iteration) is a chain of C / 4 serial load latencies.
template <int CPW>
__global__ void __launch_bounds__(256)
layernorm_c_fwd_regs_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                            const float* __restrict__ beta, float* __restrict__ out, float* __restrict__ mean,
                            float* __restrict__ rstd, int T, float eps) {
  constexpr int C = 4 * CPW;
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
  const bool ok = t < T;
  const size_t base = (size_t)b * C * T + (ok ? t : 0);
  float v[CPW];
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < CPW / 8; ++r) {
    float xv[8], yv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = base + (size_t)(wv + 4 * (8 * r + u)) * T;
      xv[u] = x[o];
      yv[u] = y ? y[o] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[8 * r + u] = xv[u] + yv[u];
      s += v[8 * r + u];
    }
  }
  red[0][wv][lane] = s;
  __syncthreads();
  const float mu = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / C;
  float q = 0.f;
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    v[k] -= mu;
    q += v[k] * v[k];
  }
  red[1][wv][lane] = q;
  __syncthreads();
  const float var = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / C;
  const float rs = rsqrtf(var + eps);
  if (ok) {
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int c = wv + 4 * k;
      out[base + (size_t)c * T] = v[k] * rs * gamma[c] + beta[c];
    }
    if (wv == 0) { mean[(size_t)b * T + t] = mu; rstd[(size_t)b * T + t] = rs; }
  }
}

// dx = rstd * (g*dy - mean_c(g*dy) - xhat * mean_c(g*dy*xhat)); dgamma += dy*xhat, dbeta += dy (atomics)
__global__ void __launch_bounds__(256)
layernorm_c_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                       const float* __restrict__ mean, const float* __restrict__ rstd,
                       const float* __restrict__ dout, float* __restrict__ dx, float* __restrict__ dgamma,
                       float* __restrict__ dbeta, int C, int T) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
  const bool ok = t < T;
  const size_t base = (size_t)b * C * T + t;
  const float mu = ok ? mean[(size_t)b * T + t] : 0.f, rs = ok ? rstd[(size_t)b * T + t] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  // eight channels per round with their loads issued together: one channel per iteration paid a global-load latency per
  // channel (the wave reductions and atomics between the loads keep the compiler from overlapping them): 172 us per launch
  // at B = 32, C = 256, T = 204
  for (int c0 = wv; c0 < C; c0 += 32) {
    float xv[8], yv[8], dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + 4 * u;
      const bool in = ok && c < C;
      xv[u] = in ? x[base + (size_t)c * T] : 0.f;
      yv[u] = (in && y) ? y[base + (size_t)c * T] : 0.f;
      dv[u] = in ? dout[base + (size_t)c * T] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = c0 + 4 * u;
      if (c >= C) break;
      const float xh = ok ? (xv[u] + yv[u] - mu) * rs : 0.f;
      const float dyv = dv[u];
      const float gd = dyv * gamma[c];
      s1 += gd; s2 += gd * xh;
      const float pg = wsum(dyv * xh), pb = wsum(dyv);
      if (lane == 0) { unsafeAtomicAdd(dgamma + c, pg); unsafeAtomicAdd(dbeta + c, pb); }
    }
  }
  red[0][wv][lane] = s1; red[1][wv][lane] = s2;
  __syncthreads();
  const float m1 = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / C;
  const float m2 = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / C;
  for (int c = wv; c < C; c += 4)
    if (ok) {
      const float xh = (x[base + (size_t)c * T] + (y ? y[base + (size_t)c * T] : 0.f) - mu) * rs;
      dx[base + (size_t)c * T] = rs * (dout[base + (size_t)c * T] * gamma[c] - m1 - xh * m2);
    }
}

// ---- banded relative-position softmax ---------------------------------------------------------------
// One wave per (g = b*H + h, query i).  scores S[g,i,j] hold (q_i/sqrt(dk)).k_j; this adds the
// relative-key logits q_i.E_k[j-i+w] on the band, applies masked_fill(mask_i*mask_j == 0, -1e4),
// the softmax over j, and writes P[g,i,j] and its transpose Pt[g,j,i] (the layout the P.V
// contraction consumes).
__global__ void __launch_bounds__(64)
rel_softmax_fwd_kernel(const float* __restrict__ S, const float* __restrict__ q, const float* __restrict__ embk,
                       const float* __restrict__ mask, float* __restrict__ P, float* __restrict__ Pd,
                       float* __restrict__ Pt, int H, int dk, int T, int w, float qscale, float pdrop,
                       unsigned long long seed, const unsigned long long* __restrict__ seed_off) {
  if (seed_off) seed += *seed_off;
  const int i = blockIdx.x, g = blockIdx.y;
  const int b = g / H;
  const int lane = threadIdx.x;
  __shared__ float rel[32];
  // rel[r] = sum_d qscale*q[g, d, i] * embk[r, d]
  const int nr = 2 * w + 1;
  for (int r = 0; r < nr; ++r) {
    float s = 0.f;
    for (int d = lane; d < dk; d += 64) s += q[((size_t)g * dk + d) * T + i] * embk[r * dk + d];
    s = wsum_all(s);
    if (lane == 0) rel[r] = s * qscale;
  }
  __syncthreads();
  const float mi = mask[(size_t)b * T + i];
  const float* Srow = S + ((size_t)g * T + i) * T;
  float* Prow = P + ((size_t)g * T + i) * T;
  float mx = -INFINITY;
  for (int j = lane; j < T; j += 64) {
    float v = Srow[j];
    const int r = j - i + w;
    if (r >= 0 && r < nr) v += rel[r];
    if (mi * mask[(size_t)b * T + j] == 0.f) v = -1e4f;
    Prow[j] = v;
    mx = fmaxf(mx, v);
  }
  mx = wmax_all(mx);
  float sum = 0.f;
  for (int j = lane; j < T; j += 64) { const float e = expf(Prow[j] - mx); Prow[j] = e; sum += e; }
  sum = wsum_all(sum);
  const float inv = 1.f / sum;
  const float inv_keep = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
  for (int j = lane; j < T; j += 64) {
    const float pv = Prow[j] * inv;
    Prow[j] = pv;
    float pd = pv;
    if (pdrop > 0.f) {
      pd = pv * drop_scale(seed, ((unsigned long long)g * T + i) * T + j, pdrop, inv_keep);
      Pd[((size_t)g * T + i) * T + j] = pd;
    }
    Pt[((size_t)g * T + j) * T + i] = pd;
  }
}

// out[g*dk + d, i] += sum_r P[g,i,i+r-w] * embv[r,d]       (relative values, band of 2w+1)
__global__ void rel_value_fwd_kernel(const float* __restrict__ P, const float* __restrict__ embv,
                                     float* __restrict__ out, int dk, int T, int w, size_t n) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int i = (int)(idx % T);
  const size_t gd = idx / T;
  const int d = (int)(gd % dk);
  const size_t g = gd / dk;
  const float* Prow = P + (g * T + i) * T;
  float s = 0.f;
  for (int r = 0; r <= 2 * w; ++r) {
    const int j = i + r - w;
    if (j >= 0 && j < T) s += Prow[j] * embv[r * dk + d];
  }
  out[idx] += s;
}

// softmax backward with the band terms.  One wave per (g, i):
//   dPfull[i,j] = dP[i,j] + (band) sum_d dO[g,d,i]*embv[j-i+w,d]
//   dS = P * (dPfull - sum_j dPfull*P), zero where masked; writes dS (in place over dP) and dSt.
//   dembv[r,d] += P[i,i+r-w]*dO[d,i];  dembk[r,d] += dS[i,i+r-w]*qscale*q[d,i];
//   dq_band[g,d,i] = qscale * sum_r dS[i,i+r-w]*embk[r,d]   (written, not accumulated)
// One workgroup (a wavefront) handles RS_ROWS query rows of one (batch, head): the gradients of the two
// relative-position tables are summed over those rows in registers and leave with ONE atomic per table entry
// (a workgroup per row hammered 2*(2w+1)*dk addresses with T*B*H-way contention: 385 us for a 1.3 M-element
// softmax).
constexpr int RS_ROWS = 16, RS_NRMAX = 16, RS_DD = 2;

__global__ void __launch_bounds__(64)
rel_softmax_bwd_kernel(const float* __restrict__ P, const float* __restrict__ Pd, float* __restrict__ dP,
                       const float* __restrict__ dO,
                       const float* __restrict__ q, const float* __restrict__ embk,
                       const float* __restrict__ embv, const float* __restrict__ mask, float* __restrict__ dSt,
                       float* __restrict__ dqband, float* __restrict__ dembk, float* __restrict__ dembv, int H,
                       int dk, int T, int w, float qscale) {
  const int g = blockIdx.y;
  const int b = g / H;
  const int lane = threadIdx.x;
  const int nr = 2 * w + 1;
  __shared__ float relv[32];
  __shared__ float dsb[32];
  float ek[RS_NRMAX][RS_DD], ev[RS_NRMAX][RS_DD];
#pragma unroll
  for (int r = 0; r < RS_NRMAX; ++r)
#pragma unroll
    for (int dd = 0; dd < RS_DD; ++dd) ek[r][dd] = ev[r][dd] = 0.f;
  const int i_lo = blockIdx.x * RS_ROWS, i_hi = i_lo + RS_ROWS < T ? i_lo + RS_ROWS : T;
  for (int i = i_lo; i < i_hi; ++i) {
    for (int r = 0; r < nr; ++r) {
      float s = 0.f;
      for (int d = lane; d < dk; d += 64) s += dO[((size_t)g * dk + d) * T + i] * embv[r * dk + d];
      s = wsum_all(s);
      if (lane == 0) relv[r] = s;
    }
    __syncthreads();
    const float* Prow = P + ((size_t)g * T + i) * T;
    const float* Pdrow = Pd + ((size_t)g * T + i) * T;  // dropped probabilities (== P when p_dropout = 0)
    float* dProw = dP + ((size_t)g * T + i) * T;
    float dot = 0.f;
    for (int j = lane; j < T; j += 64) {
      float v = dProw[j];
      const int r = j - i + w;
      if (r >= 0 && r < nr) v += relv[r];
      dProw[j] = v;
      dot += v * Pdrow[j];
    }
    dot = wsum_all(dot);
    const float mi = mask[(size_t)b * T + i];
    for (int j = lane; j < T; j += 64) {
      float ds = Pdrow[j] * dProw[j] - Prow[j] * dot;
      if (mi * mask[(size_t)b * T + j] == 0.f) ds = 0.f;
      dProw[j] = ds;
      dSt[((size_t)g * T + j) * T + i] = ds;
      const int r = j - i + w;
      if (r >= 0 && r < nr) dsb[r] = ds;
    }
    __syncthreads();
#pragma unroll
    for (int dd = 0; dd < RS_DD; ++dd) {
      const int d = lane + 64 * dd;
      if (d >= dk) continue;
      const float qv = q[((size_t)g * dk + d) * T + i] * qscale;
      const float dov = dO[((size_t)g * dk + d) * T + i];
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < RS_NRMAX; ++r) {
        const int j = i + r - w;
        if (r >= nr || j < 0 || j >= T) continue;
        acc += dsb[r] * embk[r * dk + d];
        ek[r][dd] += dsb[r] * qv;
        ev[r][dd] += Pdrow[j] * dov;
      }
      dqband[((size_t)g * dk + d) * T + i] = acc * qscale;
    }
    __syncthreads();  // relv / dsb are rewritten by the next row
  }
#pragma unroll
  for (int dd = 0; dd < RS_DD; ++dd) {
    const int d = lane + 64 * dd;
    if (d >= dk) continue;
#pragma unroll
    for (int r = 0; r < RS_NRMAX; ++r) {
      if (r >= nr) continue;
      unsafeAtomicAdd(dembk + r * dk + d, ek[r][dd]);
      unsafeAtomicAdd(dembv + r * dk + d, ev[r][dd]);
    }
  }
}

// ---- KL loss (losses.py:40-55) -------------------------------------------------------------------------
// out[0] += sum((logs_p - logs_q - 0.5 + 0.5 (z_p-m_p)^2 exp(-2 logs_p)) * mask) ; out[1] += sum(mask) once
__global__ void __launch_bounds__(256)
kl_fwd_kernel(const float* __restrict__ zp, const float* __restrict__ lq, const float* __restrict__ mp,
              const float* __restrict__ lp, const float* __restrict__ mask, float* __restrict__ out, int C, int T,
              size_t n) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int t = (int)(i % T);
    const size_t b = i / ((size_t)C * T);
    const float d = zp[i] - mp[i];
    s += (lp[i] - lq[i] - 0.5f + 0.5f * d * d * expf(-2.f * lp[i])) * mask[b * T + t];
  }
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void __launch_bounds__(256) sum_kernel(const float* __restrict__ x, float* __restrict__ out, size_t n) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i];
  s = wsum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// gradients of kl = num/den w.r.t. z_p, logs_q, m_p, logs_p ; sc[0] = gout, sc[1] = den
__global__ void kl_bwd_kernel(const float* __restrict__ zp, const float* __restrict__ mp,
                              const float* __restrict__ lp, const float* __restrict__ mask,
                              const float* __restrict__ gout, const float* __restrict__ den,
                              float* __restrict__ dzp, float* __restrict__ dlq, float* __restrict__ dmp,
                              float* __restrict__ dlp, int C, int T, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const size_t b = i / ((size_t)C * T);
  const float k = gout[0] / den[0] * mask[b * T + t];
  const float d = zp[i] - mp[i];
  const float e = expf(-2.f * lp[i]);
  dzp[i] = k * d * e;
  dmp[i] = -k * d * e;
  dlq[i] = -k;
  dlp[i] = k * (1.f - d * d * e);
}

// ---- nearest interpolation along T and segment slicing ---------------------------------------------------
// y[r, to] = x[r, floor(to * Tin / Tout)]   (F.interpolate mode="nearest")
__global__ void nearest_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Tin, int Tout, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int to = (int)(i % Tout);
  const size_t r = i / Tout;
  int ti = (int)floorf((float)to * ((float)Tin / (float)Tout));
  if (ti > Tin - 1) ti = Tin - 1;
  y[i] = x[r * Tin + ti];
}
__global__ void nearest_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Tin, int Tout, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int to = (int)(i % Tout);
  const size_t r = i / Tout;
  int ti = (int)floorf((float)to * ((float)Tin / (float)Tout));
  if (ti > Tin - 1) ti = Tin - 1;
  unsafeAtomicAdd(dx + r * Tin + ti, dy[i]);
}

// The same with the index map of OTHER sizes: raw[0] = Tin_raw <= Tin (<= 0: Tin itself), raw[1] = Tout_raw <= Tout (device
// int64 pair).  A batch
// whose padded lengths were rounded up to bucket multiples (data/collate.py: bucket_batch) keeps the alignment the
// reference's own padding gives it -- F.interpolate(m_p, size=y_spec.shape[2]) maps PADDED content frames onto PADDED
// spectrogram frames (synthesizer_svc.py:82-83), so the map depends on the batch's raw maxima, not on the bucket sizes:
//   y[r, to] = x[r, floor(to * Tin_raw / Tout_raw)] for to < Tout_raw, 0 in the bucket padding beyond.
// The sizes are read from device memory so that a recorded batch (light/graphed.py) replays with each batch's own pair.
__global__ void nearest_raw_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int Tin, int Tout,
                                       const long long* __restrict__ raw, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int tir = raw[0] > 0 ? (int)raw[0] : Tin, tor = (int)raw[1];  // (raw[0] <= 0: the content side was not re-padded)
  const int to = (int)(i % Tout);
  const size_t r = i / Tout;
  float v = 0.f;
  if (to < tor && tir > 0 && tir <= Tin) {
    int ti = (int)floorf((float)to * ((float)tir / (float)tor));
    if (ti > tir - 1) ti = tir - 1;
    v = x[r * Tin + ti];
  }
  y[i] = v;
}
__global__ void nearest_raw_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int Tin, int Tout,
                                       const long long* __restrict__ raw, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int tir = raw[0] > 0 ? (int)raw[0] : Tin, tor = (int)raw[1];  // (raw[0] <= 0: the content side was not re-padded)
  const int to = (int)(i % Tout);
  const size_t r = i / Tout;
  if (to >= tor || tir <= 0 || tir > Tin) return;
  int ti = (int)floorf((float)to * ((float)tir / (float)tor));
  if (ti > tir - 1) ti = tir - 1;
  unsafeAtomicAdd(dx + r * Tin + ti, dy[i]);
}

// y[b, c, s] = x[b, c, ids[b]*mul + s]  (0 beyond T), s < S
__global__ void slice_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ ids, int mul,
                                 float* __restrict__ y, int C, int T, int S, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = (int)(i % S);
  const size_t bc = i / S;
  const size_t b = bc / C;
  const long long t = ids[b] * mul + s;
  y[i] = (t >= 0 && t < T) ? x[bc * T + t] : 0.f;
}
__global__ void slice_bwd_kernel(const float* __restrict__ dy, const int64_t* __restrict__ ids, int mul,
                                 float* __restrict__ dx, int C, int T, int S, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int s = (int)(i % S);
  const size_t bc = i / S;
  const size_t b = bc / C;
  const long long t = ids[b] * mul + s;
  if (t >= 0 && t < T) dx[bc * T + t] = dy[i];
}

// ---- embedding rows as [B, C, T] (content_encoder.py:58-60: emb_pitch(pitch).transpose(1, -1); synthesizer_svc.py:77:
// emb_g(sid).unsqueeze(-1)) and the table gradient ------------------------------------------------------------------------
// y[b, c, t] = W[idx[b, t], c]: the rows of a 32-frame tile are read along c (coalesced), transposed through LDS and written
// along t (coalesced).  An index outside [0, rows) gives a zero column.
__global__ __launch_bounds__(256) void embedding_t_fwd_kernel(const long long* __restrict__ idx, const float* __restrict__ W,
                                                              float* __restrict__ y, int T, int C, int rows,
                                                              int* __restrict__ err) {
  extern __shared__ float tile[];  // [32][C + 1]
  const int b = blockIdx.y, t0 = blockIdx.x * 32;
  const int nt = min(32, T - t0);
  const int ld = C + 1;
  for (int e = threadIdx.x; e < nt * C; e += 256) {
    const int tt = e / C, c = e - tt * C;
    const long long r = idx[(size_t)b * T + t0 + tt];
    const bool ok = r >= 0 && r < rows;
    tile[tt * ld + c] = ok ? W[(size_t)r * C + c] : 0.f;
    // nn.Embedding fails loudly on such an index (content_encoder.py:40 / synthesizer_svc.py:68); a kernel cannot raise, so it
    // counts the offending positions in the caller's flag word (read back by the host at its next check point)
    if (!ok && c == 0 && err) atomicAdd(err, 1);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < nt * C; e += 256) {
    const int c = e / nt, tt = e - c * nt;
    y[((size_t)b * C + c) * T + t0 + tt] = tile[tt * ld + c];
  }
}

// dW[r, c] (+)= sum over the positions n = b T + t with idx[n] == r of dy[b, c, t], positions in ascending order: ONE
// workgroup owns row r, so there are no atomics and the sum order is fixed (torch's embedding_dense_backward sorts the
// indices with thrust and reads the segment count back to the host -- a sync inside the step, and on this ROCm build not
// capturable into a HIP graph: the replayed partition kernel faulted).  The table is small (512 x 128..256): every workgroup
// scans the N <= ~10^4 indices from L2 and gathers its ~N / rows matching columns.
__global__ __launch_bounds__(256) void embedding_t_bwd_kernel(const long long* __restrict__ idx, const float* __restrict__ dy,
                                                              float* __restrict__ dW, int N, int T, int C, int accumulate) {
  __shared__ int list[256];
  __shared__ int wcount[4];
  const int r = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int MAXC = 4;  // channels per thread: C <= 1024
  float acc[MAXC];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) acc[j] = 0.f;
  for (int base = 0; base < N; base += 256) {
    const int n = base + (int)threadIdx.x;
    const bool hit = n < N && idx[n] == (long long)r;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wcount[wave] = __popcll(m);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) off += wcount[w];
      total += wcount[w];
    }
    if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = n;
    __syncthreads();
    for (int k = 0; k < total; ++k) {
      const int nn = list[k];
      const int b = nn / T, t = nn - b * T;
#pragma unroll
      for (int j = 0; j < MAXC; ++j) {
        const int c = (int)threadIdx.x + 256 * j;
        if (c < C) acc[j] += dy[((size_t)b * C + c) * T + t];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = (int)threadIdx.x + 256 * j;
    if (c < C) {
      float* p = dW + (size_t)r * C + c;
      *p = accumulate ? *p + acc[j] : acc[j];
    }
  }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int vcv_wn_gate_fwd(const float* xin, const float* g, int gstride, int goff, float* acts, int B,
                               int H, int T, void* stream) {
  const size_t n = (size_t)B * H * T;
  if (!xin || !acts || n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(wn_gate_fwd_kernel, g1(n), dim3(256), 0, ST, xin, g, gstride, goff, acts, H, T, n);
  return vcv_check_launch();
}

extern "C" int vcv_wn_gate_bwd(const float* xin, const float* g, int gstride, int goff, const float* dacts,
                               float* dxin, int B, int H, int T, void* stream) {
  const size_t n = (size_t)B * H * T;
  if (!xin || !dacts || !dxin || n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(wn_gate_bwd_kernel, g1(n), dim3(256), 0, ST, xin, g, gstride, goff, dacts, dxin, H, T, n);
  return vcv_check_launch();
}

extern "C" int vcv_row_sum(const float* x, float* out, int R, int T, int inner, int ostride, int ooff,
                           void* stream) {
  if (!x || !out || R <= 0 || T <= 0 || inner <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(row_sum_kernel, dim3(R), dim3(64), 0, ST, x, out, T, inner, ostride, ooff);
  return vcv_check_launch();
}

extern "C" int vcv_wn_res_skip_fwd(const float* x, const float* out, const float* rs, const float* mask,
                                   float* xn, float* on, int B, int H, int T, int last, void* stream) {
  const size_t n = (size_t)B * H * T;
  if (!rs || !on || n == 0 || (!last && (!x || !mask || !xn))) return VCV_EINVAL;
  hipLaunchKernelGGL(wn_res_skip_fwd_kernel, g1(n), dim3(256), 0, ST, x, out, rs, mask, xn, on, H, T, last, n);
  return vcv_check_launch();
}

extern "C" int vcv_wn_res_skip_bwd(const float* dxn, const float* don, const float* mask, float* drs, float* dx,
                                   int B, int H, int T, void* stream) {
  const size_t n = (size_t)B * H * T;
  if (!mask || !drs || !dx || n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(wn_res_skip_bwd_kernel, g1(n), dim3(256), 0, ST, dxn, don, mask, drs, dx, H, T, n);
  return vcv_check_launch();
}

extern "C" int vcv_split_sample_fwd(const float* stats, const float* eps, const float* mask, float* m,
                                    float* logs, float* z, int B, int C, int T, void* stream) {
  const size_t n = (size_t)B * C * T;
  if (!stats || !mask || !m || !logs || n == 0 || (eps && !z)) return VCV_EINVAL;
  hipLaunchKernelGGL(split_sample_fwd_kernel, g1(n), dim3(256), 0, ST, stats, eps, mask, m, logs, z, C, T, n);
  return vcv_check_launch();
}

extern "C" int vcv_split_sample_bwd(const float* dm, const float* dlogs, const float* dz, const float* eps,
                                    const float* logs, const float* mask, float* dstats, int B, int C, int T,
                                    void* stream) {
  const size_t n = (size_t)B * C * T;
  if (!mask || !dstats || n == 0 || (dz && eps && !logs)) return VCV_EINVAL;
  hipLaunchKernelGGL(split_sample_bwd_kernel, g1(n), dim3(256), 0, ST, dm, dlogs, dz, eps, logs, mask, dstats,
                     C, T, n);
  return vcv_check_launch();
}

extern "C" int vcv_coupling(const float* x1, const float* m, const float* mask, float* y, int B, int C, int T,
                            int reverse, void* stream) {
  const size_t n = (size_t)B * C * T;
  if (!x1 || !m || !mask || !y || n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(coupling_kernel, g1(n), dim3(256), 0, ST, x1, m, mask, y, C, T, reverse, n);
  return vcv_check_launch();
}

extern "C" int vcv_prior_sample(const float* m, const float* logs, const float* noise, float* z, int64_t n,
                                float noise_scale, void* stream) {
  if (!m || !logs || !noise || !z || n <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(prior_sample_kernel, g1((size_t)n), dim3(256), 0, ST, m, logs, noise, z, noise_scale, (size_t)n);
  return vcv_check_launch();
}

extern "C" int vcv_layernorm_c_fwd(const float* x, const float* y, const float* gamma, const float* beta,
                                   float* out, float* mean, float* rstd, int B, int C, int T, float eps,
                                   void* stream) {
  if (!x || !gamma || !beta || !out || !mean || !rstd || B <= 0 || C <= 0 || T <= 0) return VCV_EINVAL;
  const bool regs_on = vcv_tuning().ln_regs != 0;
  if (regs_on && C == 256)
    hipLaunchKernelGGL(layernorm_c_fwd_regs_kernel<64>, dim3(vcv_cdiv(T, 64), B), dim3(256), 0, ST, x, y, gamma, beta, out, mean, rstd, T, eps);
  else if (regs_on && C == 128)
    hipLaunchKernelGGL(layernorm_c_fwd_regs_kernel<32>, dim3(vcv_cdiv(T, 64), B), dim3(256), 0, ST, x, y, gamma, beta, out, mean, rstd, T, eps);
  else
    hipLaunchKernelGGL(layernorm_c_fwd_kernel, dim3(vcv_cdiv(T, 64), B), dim3(256), 0, ST, x, y, gamma, beta, out,
                       mean, rstd, C, T, eps);
  return vcv_check_launch();
}

// The same for C = 4 * CPW (CPW = 32 / 64: both configs' hidden widths) with the wave's CPW channels of x-hat and g * dy
// kept in REGISTERS between the two passes: the generic kernel's second pass re-loaded x, y and dy one channel per iteration
// -- 64 serial global-load latencies per wave, ~100 of its 170 us at B = 32, C = 256, T = 204 -- and its 512 atomics per
// workgroup hit the same 512 words from all 128 workgroups.  Here every load of a wave is issued in rounds of 8 channels
// x 3 tensors, nothing is loaded twice, and the per-channel sums of a workgroup leave as ONE partial row per workgroup
// ([workgroup][2][C] behind no atomics; layernorm_c_bwd_reduce_kernel adds them in index order: deterministic).
template <int CPW>
__global__ void __launch_bounds__(256)
layernorm_c_bwd_regs_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ gamma,
                            const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ dout,
                            float* __restrict__ dx, float* __restrict__ part, int T) {
  constexpr int C = 4 * CPW;
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.y, t = blockIdx.x * 64 + lane;
  const bool ok = t < T;
  const size_t base = (size_t)b * C * T + (ok ? t : 0);
  const float mu = ok ? mean[(size_t)b * T + t] : 0.f, rs = ok ? rstd[(size_t)b * T + t] : 0.f;
  float xh[CPW], gd[CPW];
  float s1 = 0.f, s2 = 0.f;
  float* prow = part + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 * C;
#pragma unroll
  for (int r = 0; r < CPW / 8; ++r) {
    float xv[8], yv[8], dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const size_t o = base + (size_t)(wv + 4 * (8 * r + u)) * T;
      xv[u] = x[o];
      yv[u] = y ? y[o] : 0.f;
      dv[u] = dout[o];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int c = wv + 4 * (8 * r + u);
      const float h = ok ? (xv[u] + yv[u] - mu) * rs : 0.f;
      const float d = ok ? dv[u] : 0.f;
      const float g = d * gamma[c];
      xh[8 * r + u] = h;
      gd[8 * r + u] = g;
      s1 += g;
      s2 += g * h;
      const float pg = wsum(d * h), pb = wsum(d);
      if (lane == 0) prow[c] = pg, prow[C + c] = pb;
    }
  }
  red[0][wv][lane] = s1;
  red[1][wv][lane] = s2;
  __syncthreads();
  const float m1 = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) * (1.f / C);
  const float m2 = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) * (1.f / C);
  if (ok) {
#pragma unroll
    for (int k = 0; k < CPW; ++k) dx[base + (size_t)(wv + 4 * k) * T] = rs * (gd[k] - m1 - xh[k] * m2);
  }
}

// dgamma / dbeta [C] = sum over the workgroups' partial rows, in index order
__global__ void __launch_bounds__(256)
layernorm_c_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma, float* __restrict__ dbeta, int C, int nwg) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // element of [dgamma | dbeta]
  if (i >= 2 * C) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int q = 0;
  for (; q + 3 < nwg; q += 4) {
    a0 += part[(size_t)q * 2 * C + i];
    a1 += part[(size_t)(q + 1) * 2 * C + i];
    a2 += part[(size_t)(q + 2) * 2 * C + i];
    a3 += part[(size_t)(q + 3) * 2 * C + i];
  }
  for (; q < nwg; ++q) a0 += part[(size_t)q * 2 * C + i];
  const float v = (a0 + a1) + (a2 + a3);
  if (i < C) dgamma[i] = v; else dbeta[i - C] = v;
}

// Workspace floats vcv_layernorm_c_bwd_ws wants for (B, C, T): 0 = none (the atomics form).
extern "C" int64_t vcv_layernorm_c_bwd_scratch(int B, int C, int T) {
  const bool regs_on = vcv_tuning().ln_regs != 0;
  if (!regs_on || (C != 128 && C != 256) || B <= 0 || T <= 0) return 0;
  return (int64_t)vcv_cdiv(T, 64) * B * 2 * C;
}

// C = 128 / 256: the register-resident form; its per-workgroup partial rows go to `scratch` (>= vcv_layernorm_c_bwd_scratch
// floats, caller-owned: per call, so two devices or two streams never share it) and are consumed by the reduce launch.
extern "C" int vcv_layernorm_c_bwd_ws(const float* x, const float* y, const float* gamma, const float* mean,
                                      const float* rstd, const float* dout, float* dx, float* dgamma, float* dbeta,
                                      int B, int C, int T, float* scratch, int64_t scratch_floats, void* stream) {
  if (!x || !gamma || !mean || !rstd || !dout || !dx || !dgamma || !dbeta || B <= 0 || C <= 0 || T <= 0)
    return VCV_EINVAL;
  const int64_t need = vcv_layernorm_c_bwd_scratch(B, C, T);
  if (need > 0 && scratch && scratch_floats >= need) {
    const int nwg = vcv_cdiv(T, 64) * B;
    if (C == 256)
      hipLaunchKernelGGL(layernorm_c_bwd_regs_kernel<64>, dim3(vcv_cdiv(T, 64), B), dim3(256), 0, ST, x, y, gamma, mean, rstd, dout, dx, scratch, T);
    else
      hipLaunchKernelGGL(layernorm_c_bwd_regs_kernel<32>, dim3(vcv_cdiv(T, 64), B), dim3(256), 0, ST, x, y, gamma, mean, rstd, dout, dx, scratch, T);
    hipLaunchKernelGGL(layernorm_c_bwd_reduce_kernel, dim3(vcv_cdiv(2 * C, 256)), dim3(256), 0, ST, (const float*)scratch, dgamma, dbeta, C, nwg);
    return vcv_check_launch();
  }
  if (vcv_zero_async(dgamma, sizeof(float) * C, ST) != hipSuccess) return VCV_EHIP;
  if (vcv_zero_async(dbeta, sizeof(float) * C, ST) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(layernorm_c_bwd_kernel, dim3(vcv_cdiv(T, 64), B), dim3(256), 0, ST, x, y, gamma, mean, rstd,
                     dout, dx, dgamma, dbeta, C, T);
  return vcv_check_launch();
}

// The same without a caller workspace (kept for ABI users of rounds 3-4): the register-resident form's partial rows live in a
// scratch keyed by the current DEVICE (grown on demand, old buffers kept: an enqueued reduce may still read them).  Launches
// of one device on different streams would share it -- callers that use several streams pass their own (_ws above).
extern "C" int vcv_layernorm_c_bwd(const float* x, const float* y, const float* gamma, const float* mean,
                                   const float* rstd, const float* dout, float* dx, float* dgamma, float* dbeta,
                                   int B, int C, int T, void* stream) {
  const int64_t need = vcv_layernorm_c_bwd_scratch(B, C, T);
  float* ws = nullptr;
  if (need > 0) {
    static float* scratch[16] = {};
    static int64_t floats[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return VCV_EHIP;
    if (need > floats[dev]) {
      float* nb = nullptr;
      if (hipMalloc((void**)&nb, (size_t)need * 2 * sizeof(float)) != hipSuccess) return VCV_EHIP;
      scratch[dev] = nb;
      floats[dev] = need * 2;
    }
    ws = scratch[dev];
  }
  return vcv_layernorm_c_bwd_ws(x, y, gamma, mean, rstd, dout, dx, dgamma, dbeta, B, C, T, ws, need, stream);
}

extern "C" int vcv_rel_softmax_fwd(const float* S, const float* q, const float* embk, const float* mask,
                                   float* P, float* Pd, float* Pt, int B, int H, int dk, int T, int w,
                                   float qscale, float pdrop, uint64_t seed, void* stream) {
  if (!S || !q || !embk || !mask || !P || !Pt || B <= 0 || H <= 0 || dk <= 0 || T <= 0 || w < 0 || 2 * w + 1 > 32)
    return VCV_EINVAL;
  if (pdrop < 0.f || pdrop >= 1.f || (pdrop > 0.f && !Pd)) return VCV_EINVAL;
  hipLaunchKernelGGL(rel_softmax_fwd_kernel, dim3(T, B * H), dim3(64), 0, ST, S, q, embk, mask, P, Pd, Pt, H, dk,
                     T, w, qscale, pdrop, (unsigned long long)seed, (const unsigned long long*)vcv_get_seed_offset_ptr());
  return vcv_check_launch();
}

extern "C" int vcv_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, void* stream) {
  if (!x || !y || n <= 0 || p < 0.f || p >= 1.f) return VCV_EINVAL;
  hipLaunchKernelGGL(dropout_kernel, g1((size_t)n), dim3(256), 0, ST, x, y, (size_t)n, p, 1.f / (1.f - p),
                     (unsigned long long)seed, (const unsigned long long*)vcv_get_seed_offset_ptr());
  return vcv_check_launch();
}

extern "C" int vcv_rel_value_fwd(const float* P, const float* embv, float* out, int B, int H, int dk, int T,
                                 int w, void* stream) {
  const size_t n = (size_t)B * H * dk * T;
  if (!P || !embv || !out || n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(rel_value_fwd_kernel, g1(n), dim3(256), 0, ST, P, embv, out, dk, T, w, n);
  return vcv_check_launch();
}

extern "C" int vcv_rel_softmax_bwd(const float* P, const float* Pd, float* dP, const float* dO, const float* q,
                                   const float* embk, const float* embv, const float* mask, float* dSt,
                                   float* dqband, float* dembk, float* dembv, int B, int H, int dk, int T, int w,
                                   float qscale, void* stream) {
  if (!P || !Pd || !dP || !dO || !q || !embk || !embv || !mask || !dSt || !dqband || !dembk || !dembv || 2 * w + 1 > 32)
    return VCV_EINVAL;
  const size_t ne = sizeof(float) * (2 * w + 1) * dk;
  if (vcv_zero_async(dembk, ne, ST) != hipSuccess) return VCV_EHIP;
  if (vcv_zero_async(dembv, ne, ST) != hipSuccess) return VCV_EHIP;
  if (2 * w + 1 > RS_NRMAX || dk > 64 * RS_DD) return VCV_EINVAL;
  hipLaunchKernelGGL(rel_softmax_bwd_kernel, dim3(vcv_cdiv(T, RS_ROWS), B * H), dim3(64), 0, ST, P, Pd, dP, dO, q, embk, embv, mask, dSt,
                     dqband, dembk, dembv, H, dk, T, w, qscale);
  return vcv_check_launch();
}

extern "C" int vcv_kl_fwd(const float* zp, const float* lq, const float* mp, const float* lp, const float* mask,
                          float* out2, int B, int C, int T, void* stream) {
  const size_t n = (size_t)B * C * T;
  if (!zp || !lq || !mp || !lp || !mask || !out2 || n == 0) return VCV_EINVAL;
  if (vcv_zero_async(out2, 2 * sizeof(float), ST) != hipSuccess) return VCV_EHIP;
  size_t nb = (n + 2047) / 2048;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, ST, zp, lq, mp, lp, mask, out2, C, T, n);
  const size_t nm = (size_t)B * T;
  size_t nb2 = (nm + 2047) / 2048;
  hipLaunchKernelGGL(sum_kernel, dim3((unsigned)nb2), dim3(256), 0, ST, mask, out2 + 1, nm);
  return vcv_check_launch();
}

extern "C" int vcv_kl_bwd(const float* zp, const float* mp, const float* lp, const float* mask, const float* gout,
                          const float* den, float* dzp, float* dlq, float* dmp, float* dlp, int B, int C, int T,
                          void* stream) {
  const size_t n = (size_t)B * C * T;
  if (!zp || !mp || !lp || !mask || !gout || !den || !dzp || !dlq || !dmp || !dlp || n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(kl_bwd_kernel, g1(n), dim3(256), 0, ST, zp, mp, lp, mask, gout, den, dzp, dlq, dmp, dlp, C, T,
                     n);
  return vcv_check_launch();
}

extern "C" int vcv_nearest_fwd(const float* x, float* y, int R, int Tin, int Tout, void* stream) {
  const size_t n = (size_t)R * Tout;
  if (!x || !y || n == 0 || Tin <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(nearest_fwd_kernel, g1(n), dim3(256), 0, ST, x, y, Tin, Tout, n);
  return vcv_check_launch();
}

extern "C" int vcv_nearest_bwd(const float* dy, float* dx, int R, int Tin, int Tout, void* stream) {
  const size_t n = (size_t)R * Tout;
  if (!dy || !dx || n == 0 || Tin <= 0) return VCV_EINVAL;
  if (vcv_zero_async(dx, sizeof(float) * (size_t)R * Tin, ST) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(nearest_bwd_kernel, g1(n), dim3(256), 0, ST, dy, dx, Tin, Tout, n);
  return vcv_check_launch();
}

extern "C" int vcv_nearest_raw_fwd(const float* x, float* y, int R, int Tin, int Tout, const void* raw, void* stream) {
  const size_t n = (size_t)R * Tout;
  if (!x || !y || !raw || n == 0 || Tin <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(nearest_raw_fwd_kernel, g1(n), dim3(256), 0, ST, x, y, Tin, Tout, (const long long*)raw, n);
  return vcv_check_launch();
}

extern "C" int vcv_nearest_raw_bwd(const float* dy, float* dx, int R, int Tin, int Tout, const void* raw, void* stream) {
  const size_t n = (size_t)R * Tout;
  if (!dy || !dx || !raw || n == 0 || Tin <= 0) return VCV_EINVAL;
  if (vcv_zero_async(dx, sizeof(float) * (size_t)R * Tin, ST) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(nearest_raw_bwd_kernel, g1(n), dim3(256), 0, ST, dy, dx, Tin, Tout, (const long long*)raw, n);
  return vcv_check_launch();
}

extern "C" int vcv_slice_fwd(const float* x, const int64_t* ids, int mul, float* y, int B, int C, int T, int S,
                             void* stream) {
  const size_t n = (size_t)B * C * S;
  if (!x || !ids || !y || n == 0) return VCV_EINVAL;
  hipLaunchKernelGGL(slice_fwd_kernel, g1(n), dim3(256), 0, ST, x, ids, mul, y, C, T, S, n);
  return vcv_check_launch();
}

extern "C" int vcv_slice_bwd(const float* dy, const int64_t* ids, int mul, float* dx, int B, int C, int T, int S,
                             void* stream) {
  const size_t n = (size_t)B * C * S;
  if (!dy || !ids || !dx || n == 0) return VCV_EINVAL;
  if (vcv_zero_async(dx, sizeof(float) * (size_t)B * C * T, ST) != hipSuccess) return VCV_EHIP;
  hipLaunchKernelGGL(slice_bwd_kernel, g1(n), dim3(256), 0, ST, dy, ids, mul, dx, C, T, S, n);
  return vcv_check_launch();
}

extern "C" int vcv_embedding_t_fwd_checked(const void* idx, const float* W, float* y, int B, int T, int C, int rows, int* err,
                                           void* stream) {
  if (!idx || !W || !y || B <= 0 || T <= 0 || C <= 0 || rows <= 0) return VCV_EINVAL;
  const size_t lds = sizeof(float) * 32 * (size_t)(C + 1);
  if (lds > 64 * 1024) return VCV_EINVAL;
  hipLaunchKernelGGL(embedding_t_fwd_kernel, dim3(vcv_cdiv(T, 32), B), dim3(256), lds, ST, (const long long*)idx, W, y, T, C, rows,
                     err);
  return vcv_check_launch();
}

extern "C" int vcv_embedding_t_fwd(const void* idx, const float* W, float* y, int B, int T, int C, int rows, void* stream) {
  return vcv_embedding_t_fwd_checked(idx, W, y, B, T, C, rows, nullptr, stream);
}

extern "C" int vcv_embedding_t_bwd(const void* idx, const float* dy, float* dW, int B, int T, int C, int rows, int accumulate,
                                   void* stream) {
  if (!idx || !dy || !dW || B <= 0 || T <= 0 || C <= 0 || C > 1024 || rows <= 0) return VCV_EINVAL;
  hipLaunchKernelGGL(embedding_t_bwd_kernel, dim3(rows), dim3(256), 0, ST, (const long long*)idx, dy, dW, B * T, T, C, accumulate);
  return vcv_check_launch();
}

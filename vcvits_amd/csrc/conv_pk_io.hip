// conv_pk_io.hip -- the bf16 packed-operand convolution kernel (conv_pk_kernel.h, design notes in conv_pk.hip) with bf16
// ACTIVATIONS in HBM: `x`, `y`, `res` and the accumulate target are bf16 tensors (VcvConvArgs.io == VCV_IO_BF16).
//
// Reference: the HiFi-GAN decoder under fp16 autocast (synthesizer_svc.py:108 behind train.py:104-106 `precision=16`;
// infer.py:84): AMP stores every conv <-> conv activation in half precision.  With fp32 activations the 48 kHz decode
// of 64 x 10 s moved 5.7 GB per conv launch at 4 TB/s with the matrix pipe at 0.12 of its peak (profiles/r3_48k_infer_*):
// the 32- and 64-channel stages are HBM-bound, so halving the bytes is what moves them.
//   * input: 16-byte buffer loads of EIGHT consecutive positions per channel (4-byte aligned rows: an even number of
//     elements per row), transposed to the channel-innermost LDS image by register naming; no conversion at all unless
//     the input leaky-ReLU is fused (then bf16 -> fp32 -> leaky -> bf16: one rounding, the one the fp32-activation path
//     applies to its operand);
//   * epilogue in fp32 (bias, activation, residual, mask, post-scale, accumulate), ONE rounding to bf16 (nearest even),
//     16-byte stores of eight columns through the wave-private LDS tile;
//   * vcv_conv_m1_bf16in_fwd: the 32 -> 1 conv_post + tanh over a bf16 input (an HBM read stream: 16-byte loads, fp32
//     FMAs, fp32 output);  vcv_cast_*: the conversions at the two ends of a bf16 chain.
// Numerics (tests/test_bf16_io_gpu.py): a launch equals the fp32 CPU convolution of the same bf16 inputs to 1e-5 before
// the output rounding; the decoder's waveform stays within the north_star's 1e-3 RMS of the fp32 oracle.
#include "conv_pk_io_inst.h"

namespace {

template <int KIND>
__global__ void __launch_bounds__(256) cast_f32_16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, size_t n) {
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i), b = *reinterpret_cast<const f32x4*>(x + i + 4);
    us8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = f32_to_us<KIND>(a[j]);
      o[j + 4] = f32_to_us<KIND>(b[j]);
    }
    *reinterpret_cast<us8*>(y + i) = o;
  } else {
    for (size_t j = i; j < n; ++j) y[j] = f32_to_us<KIND>(x[j]);
  }
}

__global__ void __launch_bounds__(256) cast_16_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y, size_t n, int kind) {
  typedef unsigned short us8 __attribute__((ext_vector_type(8)));
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const us8 a = *reinterpret_cast<const us8*>(x + i);
    f32x4 o0, o1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o0[j] = us_to_f32(a[j], kind);
      o1[j] = us_to_f32(a[j + 4], kind);
    }
    *reinterpret_cast<f32x4*>(y + i) = o0;
    *reinterpret_cast<f32x4*>(y + i + 4) = o1;
  } else {
    for (size_t j = i; j < n; ++j) y[j] = us_to_f32(x[j], kind);
  }
}

// One output channel, stride 1, (K - 1) * dil <= 7: thread = eight consecutive output positions.  Per channel a thread
// reads the 16 inputs [lo & ~1, lo & ~1 + 16) that cover its window with two 16-byte buffer loads (range-checked
// descriptors: the zero padding at both row ends costs no predicate; rows are 4-byte aligned), converts, applies the
// input leaky-ReLU and runs 8 x K fp32 FMAs out of registers; the weights are wave-uniform (scalar loads).
// grid: (ceil(T / 2048), B)
template <int K>
__global__ void __launch_bounds__(256)
conv_m1_x16_kernel(const unsigned short* __restrict__ x, int kind, const float* __restrict__ w, const float* __restrict__ bias,
                   float* __restrict__ y, int C, int Tin, int Tout, int pad, int in_leaky, int out_act, float slope) {
  const int b = blockIdx.y;
  const int u = (blockIdx.x * 256 + threadIdx.x) * 8;
  const int lo = u - pad, start = lo & ~1, sh = lo - start;  // (arithmetic: -3 & ~1 = -4)
  unsigned voff = (unsigned)start * 2u;                        // negative -> wraps -> out of range -> zeros
  asm volatile("" : "+v"(voff));
  unsigned voff1 = voff + 16u;
  asm volatile("" : "+v"(voff1));
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float sl = in_leaky ? slope : 1.f;
  const unsigned shbits = 16u * (unsigned)sh;
#pragma unroll 4
  for (int c = 0; c < C; ++c) {
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + ((size_t)b * C + c) * (size_t)Tin), 0, Tin * 2, 0x00020000);
    f32x4 q0;
    if (start < 0) {
      // the row's first thread: a 16-byte load that begins before the buffer comes back as zeros as a whole, so the window's
      // leading padding is taken dword by dword (the out-of-range ones read zero, the others their data)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (whole offset in one register: a negative register part plus a positive immediate that sum to 0 or 4 reads zero)
        unsigned vo = voff + 4u * i;
        asm volatile("" : "+v"(vo));
        q0[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, vo, 0, 0));
      }
    } else {
      q0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0));
    }
    const f32x4 q1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff1, 0, 0));
    // input element lo + i = start + sh + i, sh in {0, 1}: a funnel shift of the raw dwords by sh bf16 elements (a select
    // between neighbouring array elements becomes a run-time index, and the compiler then moves the array to LDS)
    const unsigned r[8] = {__float_as_uint(q0[0]), __float_as_uint(q0[1]), __float_as_uint(q0[2]), __float_as_uint(q0[3]),
                           __float_as_uint(q1[0]), __float_as_uint(q1[1]), __float_as_uint(q1[2]), __float_as_uint(q1[3])};
    float g[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned d = __builtin_amdgcn_alignbit(i < 7 ? r[i + 1] : 0u, r[i], shbits);
      const float t0 = us_to_f32((unsigned short)(d & 0xffffu), kind), t1 = us_to_f32((unsigned short)(d >> 16), kind);
      g[2 * i] = fmaxf(t0, t0 * sl);  // (sl = 1 without the input leaky-ReLU)
      g[2 * i + 1] = fmaxf(t1, t1 * sl);
    }
    const float* wr = w + (size_t)c * K;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const float wk = wr[k];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] += wk * g[j + k];  // (dilation 1: the launcher takes no other)
    }
  }
  if (u >= Tout) return;
  const float bv = bias ? bias[0] : 0.f;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = vcv_act(acc[j] + bv, out_act, slope);
  float* yr = y + (size_t)b * Tout + u;
  if (u + 8 <= Tout && (Tout & 3) == 0) {
    *reinterpret_cast<f32x4*>(yr) = f32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<f32x4*>(yr + 4) = f32x4{o[4], o[5], o[6], o[7]};
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (u + j < Tout) yr[j] = o[j];
  }
}

}  // namespace

// storage combinations instantiated (conv_pk_io{3,7,11,15}.hip): all-bf16 (3), fp16 x -> bf16 y (7: the first conv of a
// ResBlock pair, which stores leaky(xt) as the bf16 operand its only consumer feeds the matrix cores), bf16 x -> fp16 y
// and res (11: the second conv), fp16 -> fp16 (15: the transposed convs)
extern "C" int vcv_conv_bf16io_plan(const VcvConvArgs* args, int flip, int64_t* out) {
  if (!args) return VCV_EINVAL;
  switch (args->io) {
    case 3: return vcv_conv_io_plan_3(args, flip, out);
    case 7: return vcv_conv_io_plan_7(args, flip, out);
    case 11: return vcv_conv_io_plan_11(args, flip, out);
    case 15: return vcv_conv_io_plan_15(args, flip, out);
    default: return VCV_EINVAL;
  }
}
extern "C" int vcv_conv_bf16io_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                                   void* stream) {
  if (!args) return VCV_EINVAL;
  switch (args->io) {
    case 3: return vcv_conv_io_run_3(args, pack_ws, scratch_ws, flip, pack_valid, stream);
    case 7: return vcv_conv_io_run_7(args, pack_ws, scratch_ws, flip, pack_valid, stream);
    case 11: return vcv_conv_io_run_11(args, pack_ws, scratch_ws, flip, pack_valid, stream);
    case 15: return vcv_conv_io_run_15(args, pack_ws, scratch_ws, flip, pack_valid, stream);
    default: return VCV_EINVAL;
  }
}

// kind: 1 = bf16, 2 = fp16 (the storage kinds of conv_tile.h)
extern "C" int vcv_cast_f32_x16(const float* x, void* y, int64_t n, int kind, void* stream) {
  if (!x || !y || n <= 0 || (kind != 1 && kind != 2) || (((uintptr_t)x | (uintptr_t)y) & 15)) return VCV_EINVAL;
  const dim3 grid((unsigned)((n + 2047) / 2048)), block(256);
  if (kind == 2) hipLaunchKernelGGL(cast_f32_16_kernel<2>, grid, block, 0, (hipStream_t)stream, x, (unsigned short*)y, (size_t)n);
  else hipLaunchKernelGGL(cast_f32_16_kernel<1>, grid, block, 0, (hipStream_t)stream, x, (unsigned short*)y, (size_t)n);
  return vcv_check_launch();
}

extern "C" int vcv_cast_x16_f32(const void* x, float* y, int64_t n, int kind, void* stream) {
  if (!x || !y || n <= 0 || (kind != 1 && kind != 2) || (((uintptr_t)x | (uintptr_t)y) & 15)) return VCV_EINVAL;
  hipLaunchKernelGGL(cast_16_f32_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned short*)x, y, (size_t)n, kind);
  return vcv_check_launch();
}

extern "C" int vcv_conv_m1_x16_fwd(const void* x, int kind, const float* w, const float* bias, float* y, int B, int C, int Tin,
                                   int Tout, int K, int dil, int pad, int in_leaky, int out_act, float slope, void* stream) {
  if (!x || !w || !y || B <= 0 || C <= 0 || Tin <= 0 || Tout <= 0 || (kind != 1 && kind != 2)) return VCV_EINVAL;
  // rows of an even number of elements (4-byte aligned), window of 8 + (K - 1) * dil + 1 <= 16 elements, dilation 1
  if ((Tin & 1) || dil != 1 || pad < 0 || Tout != Tin + 2 * pad - dil * (K - 1) || (long long)Tin * 2 >= (1ll << 31)) return VCV_EINVAL;
  dim3 grid((unsigned)vcv_cdiv(Tout, 2048), (unsigned)B), block(256);
  hipStream_t st = (hipStream_t)stream;
  const unsigned short* xs = (const unsigned short*)x;
  switch (K) {
    case 3: hipLaunchKernelGGL(conv_m1_x16_kernel<3>, grid, block, 0, st, xs, kind, w, bias, y, C, Tin, Tout, pad, in_leaky, out_act, slope); break;
    case 5: hipLaunchKernelGGL(conv_m1_x16_kernel<5>, grid, block, 0, st, xs, kind, w, bias, y, C, Tin, Tout, pad, in_leaky, out_act, slope); break;
    case 7: hipLaunchKernelGGL(conv_m1_x16_kernel<7>, grid, block, 0, st, xs, kind, w, bias, y, C, Tin, Tout, pad, in_leaky, out_act, slope); break;
    default: return VCV_EINVAL;
  }
  return vcv_check_launch();
}

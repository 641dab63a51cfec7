// conv_pk.hip -- channel-innermost packed-operand implicit-GEMM convolution (forward-type and phased launches, the
// same launch family conv_dma.hip covers), in two element types:
//   bf16 (vcv_conv_bf16_*): v_mfma_f32_32x32x16_bf16, fp32 accumulate, fp32 activations in HBM;
//   fp32 (vcv_conv_pk_*):   the same kernel on 4-channel 16-byte groups and v_mfma_f32_32x32x2_f32 (exact fp32): one
//                           16-byte LDS read per operand feeds FOUR MFMAs, a quarter of conv_dma.hip's LDS read
//                           instructions and four times its MFMA work per LDS round trip -- 112-121 vs 98-103 TFLOP/s
//                           on the 1024-channel period-discriminator layers, 0.59 vs 0.49 of the fp32 peak in the step.
//
// The reference trains under AMP (configs/base.json:18, train.py:104-106: fp16 autocast of the convs, fp32 master
// weights, fp32 losses).  Here the same recipe with bf16: operands are rounded to bf16 (round-to-nearest-even,
// v_cvt_pk_bf16_f32) on their way into the matrix cores, products are exact, sums are fp32, and everything outside
// the GEMMs (activations in HBM, epilogue, losses, optimizer) stays fp32.
//
// MFMA mapping: one instruction = 32 output channels x 32 positions x 16 REDUCTION CHANNELS of one tap.  Lane
// (r = lane & 31, h = lane >> 5) holds A[m = r][c = 8h .. 8h+7] and B[c = 8h .. 8h+7][position r]: 8 consecutive
// channels = one 16-byte LDS read, so both images are channel-innermost:
//   As  [tap j][16-channel group cg][h][m (BM)][8 ch]   bf16   (packed in HBM in exactly this order: one chunk of one
//                                                              tile is a contiguous slab, copied by global_load_lds
//                                                              1 KiB per wave-instruction, as conv_dma.hip does)
//   Xs  [cg][h][position (span)][8 ch]                  bf16   (register-staged: a lane loads the 8 channels of ONE
//                                                              position -- 8 coalesced 256-byte wave loads -- or, in
//                                                              the X4 variants, of FOUR consecutive positions with one
//                                                              16-byte load per channel; applies the input leaky-ReLU,
//                                                              converts and writes 16 bytes per position: the transpose
//                                                              [c][t] -> [t][c] is register naming)
// A tap shift is a position offset of the B read: every tap re-reads the same staged span (an input element is
// fetched once per workgroup, not once per tap).  Reads: lanes r = 0..31 of one half read 512 contiguous bytes
// (stride-1 layers) or 16-byte slots 3 apart (the period discriminators' stride 3: coprime with the 16 slots of a
// bank row) -> conflict-free ds_read_b128.
// Two LDS buffers: the weight DMA and the input loads of chunk c+1 are issued before the MFMA loop of chunk c, the
// converted inputs are written after it, one barrier per chunk.  The warp-specialised variants (NP > 0) give the whole
// staging of chunk c+1 to NP producer waves instead (see the kernel's comment and wgrad_dma.hip).
#include "conv_pk_kernel.h"

extern "C" int vcv_conv_pk_pack_job(const VcvConvArgs* args, int flip, VcvPackJob* out) { return pack_job_t<F32El>(args, flip, out); }
extern "C" int vcv_conv_bf16_pack_job(const VcvConvArgs* args, int flip, VcvPackJob* out) { return pack_job_t<Bf16El>(args, flip, out); }

// Same calling convention as vcv_conv_dma_plan / vcv_conv_dma_run (include/vcvits_hip.h): out[0] = BYTES / 4 of the
// packed-weight buffer (so callers allocate it as out[0] fp32 words), out[1] = floats of per-launch scratch, out[2] =
// signature of the pack layout.
extern "C" int vcv_conv_bf16_plan(const VcvConvArgs* args, int flip, int64_t* out) { return plan_t<Bf16El>(args, flip, out); }
extern "C" int vcv_conv_bf16_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                                 void* stream) {
  return run_t<Bf16El>(args, pack_ws, scratch_ws, flip, pack_valid, stream);
}
// The same kernel on fp32 elements (4-channel 16-byte groups, v_mfma_f32_32x32x2_f32: exact fp32)
extern "C" int vcv_conv_pk_plan(const VcvConvArgs* args, int flip, int64_t* out) { return plan_t<F32El>(args, flip, out); }
extern "C" int vcv_conv_pk_run(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                               void* stream) {
  return run_t<F32El>(args, pack_ws, scratch_ws, flip, pack_valid, stream);
}

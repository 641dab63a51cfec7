// tuning.h -- ONE process-wide table of the library's A/B switches and tuning probes.
//
// Rounds 1-5 grew 39 `getenv("VCVITS_...")` calls across the kernel files, each read once into a function-local static.  They
// are folded into this struct: every field has its measured default baked in, the whole table is initialised once from ONE
// environment variable
//     VCVITS_TUNING="key=value,key=value,..."          (e.g. VCVITS_TUNING="pk_x4=0,wgrad_tile=3")
// and can be read / written at run time through the C ABI (vcv_tuning_get / vcv_tuning_set: tools/*_sweep.py, tests), so an
// A/B run no longer needs a fresh process per setting.  VCVITS_DETERMINISTIC=1 (documented in README) is the only other
// variable the library reads.  The keys are the field names.
#pragma once

struct VcvTuning {
  // ---- packed-weight convolution kernels (conv_pk_kernel.h, conv_x3.hip) ----
  int xcd_remap = 1;       // blockIdx -> tile re-deal so that the m-tiles of one column tile share an XCD's L2
  int pk_ws = 1;           // warp-specialised twins of the wide fp32 tiles (8 or 4 MFMA waves + 4 producer waves)
  int pk_ws_bf16 = 0;      // ... for the bf16-operand launches too (measured: no gain)
  int pk_x4 = 1;           // 16-byte input loads (four positions per lane and channel)
  int pk_vec = 1;          // 16-byte epilogue through a wave-private LDS tile
  int x3_variant = -1;     // split-operand kernel: fix the tile variant 0..6 (-1: the planner's choice)
  int x3_v6 = 1;           // 64 x 512 tile for the generator's 64-channel layers
  int x3_js2 = 1;          // two taps per stage where the doubled weight ring fits
  int x3_old_ks = 0;       // round-3 rule for the channel-group split
  int x3_all = 0;          // take every eligible launch, not only the shapes where the split kernel is ahead
  int x3_terms = 6;        // bf16 product terms per fp32 product: 6 or 9
  // ---- weight gradients (wgrad_dma.hip, wgrad_bf16.hip, conv_wgrad.hip) ----
  int wgrad_dma = 1;       // the LDS-DMA weight-gradient kernel (0: the register-staged one)
  int wgrad_tile = -1;     // wgrad_dma: fix the tile candidate (-1: the cost model)
  int wgrad_verbose = 0;   // wgrad_dma: print every plan
  int wgrad_bf16_ws = 1;   // producer waves in the bf16-operand weight gradient
  int wgrad_finish_vec = 1;  // 16-byte slab reads in the finishing pass
  int bias_rows = 1;       // one workgroup per channel for short rows in vcv_bias_grad
  // ---- thin / streaming kernels ----
  int c1_chunk = 0;        // conv_c1: output channels per workgroup (0: halve until the grid fills)
  int m1_lds = 1;          // conv_m1: LDS-staged window for stride-1 launches
  int c1_wgrad_pairs = 1;  // one-input-channel weight gradient by (channel, tap) pairs
  int thin_wgrad_wgs = 2048;  // thin weight gradient: workgroups aimed at
  int act_grad_vec = 1;    // float4 activation-derivative passes
  int ln_regs = 1;         // register-resident LayerNorm kernels (C = 128 / 256)
  int stft_wave = 1;       // one wavefront per STFT frame (0: the 256-threads-per-frame radix-2 form)
  int attn_rows = 1;       // fused attention: one wave per query-row block for T <= 256
  int zero_memset = 0;     // hipMemsetAsync instead of the fill kernel
  // ---- weight packs (conv_pack.hip) ----
  int pack_tile = 1;       // LDS-transposed pack kernels
  int pack_tile_bf16 = 1;
  // ---- fused ResBlock pair (resblock_pair.hip) ----
  int pair_dbg = 0;
  int pair_grid = 0;       // fix the persistent grid (0: one workgroup per CU slot)
  int pair_stream = 1;     // 64 channels x K >= 7: weights streamed tap by tap (0: leave those to two launches)
  // ---- arithmetic ----
  int deterministic = 0;   // bit-reproducible reductions (VCVITS_DETERMINISTIC=1 / vcv_set_deterministic)
};

// (defined in version.hip)
VcvTuning& vcv_tuning();
extern "C" int vcv_tuning_set(const char* key, int value);
extern "C" int vcv_tuning_get(const char* key, int* value);

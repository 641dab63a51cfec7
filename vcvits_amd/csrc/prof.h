// prof.h -- optional per-launch HIP-event timing of the MFMA kernels (used by bench.py for the
// roofline line: achieved FLOP/s of the dominant kernel measured on the launch stream itself).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#define VCV_PROF_CONV 0       /* conv_gemm_kernel */
#define VCV_PROF_WGRAD 1      /* conv_wgrad_kernel */
#define VCV_PROF_CONV_DMA 2   /* conv_dma_kernel */
#define VCV_PROF_WGRAD_DMA 3  /* wgrad_dma_kernel */
#define VCV_PROF_ATTN 4       /* fused attention kernels (attention.hip) */
#define VCV_PROF_NCLS 5

// returns a slot (>= 0) when profiling is on and an event pair was recorded before the launch
int vcv_prof_start(int cls, double flops, hipStream_t st, const int* tag = nullptr, int ntag = 0);
void vcv_prof_stop(int slot, hipStream_t st);
// Dispatch-attached timing: reserves a slot and returns its two events for hipExtLaunchKernelGGL (the kernel's own
// start / completion timestamps: no marker packets in the queue, so profiling does not serialise the stream).
// Both events are null when profiling is off.
// `bytes`: algorithmic HBM bytes of the launch (operands read once + result written once), for the traffic line.
// `roof_s`: the launch's time at the dense peak of the matrix pipe it runs on -- MFMA flops it EXECUTES / that pipe's peak
// (fp32-input MFMA: flops / 157.3e12, the default when 0; bf16 MFMA: flops / 2.5e15; split-operand fp32 launches execute
// NTERM bf16 products per fp32 product: NTERM * flops / 2.5e15).  bench.py reports sum(roof_s) / sum(measured) as the
// roofline fraction of a class that mixes pipes.
void vcv_prof_events(int cls, double flops, const int* tag, int ntag, hipEvent_t* start, hipEvent_t* stop, double bytes = 0.0,
                     double roof_s = 0.0);
#define VCV_PEAK_F32_MFMA 157.3e12
#define VCV_PEAK_BF16_MFMA 2.5e15

// Launch with dispatch-attached events when the profiler handed some out, as a plain launch otherwise (plain launches are
// what HIP-graph stream capture records; the profiler is off while a sequence is captured or replayed).
#define VCV_LAUNCH_EV(kern, grid, block, lds, st, ev0, ev1, ...)                                  \
  do {                                                                                           \
    if (ev0) hipExtLaunchKernelGGL(kern, grid, block, lds, st, ev0, ev1, 0, __VA_ARGS__);        \
    else hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                            \
  } while (0)

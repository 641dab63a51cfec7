// conv_pk_io_inst.h -- one translation unit per storage combination of the 16-bit-activation convolution kernel
// (conv_pk_kernel.h with IO != 0): #define VCV_IO_INST <io> and include this file.  The entry points below are internal
// (hidden visibility); the public vcv_conv_bf16io_plan / vcv_conv_bf16io_run (conv_pk_io.hip) dispatch on VcvConvArgs.io.
#pragma once
#include "conv_pk_kernel.h"

#define VCV_IO_CAT2(a, b) a##b
#define VCV_IO_CAT(a, b) VCV_IO_CAT2(a, b)
#define VCV_IO_DECL(io)                                                                                              \
  __attribute__((visibility("hidden"))) int VCV_IO_CAT(vcv_conv_io_plan_, io)(const VcvConvArgs*, int, int64_t*);   \
  __attribute__((visibility("hidden"))) int VCV_IO_CAT(vcv_conv_io_run_, io)(const VcvConvArgs*, float*, float*, int, int, void*);
VCV_IO_DECL(3)
VCV_IO_DECL(7)
VCV_IO_DECL(11)
VCV_IO_DECL(15)

#ifdef VCV_IO_INST
int VCV_IO_CAT(vcv_conv_io_plan_, VCV_IO_INST)(const VcvConvArgs* args, int flip, int64_t* out) {
  return plan_t<Bf16El, VCV_IO_INST>(args, flip, out);
}
int VCV_IO_CAT(vcv_conv_io_run_, VCV_IO_INST)(const VcvConvArgs* args, float* pack_ws, float* scratch_ws, int flip, int pack_valid,
                                               void* stream) {
  return run_t<Bf16El, VCV_IO_INST>(args, pack_ws, scratch_ws, flip, pack_valid, stream);
}
#endif

// conv_pk_io7.hip -- the 16-bit-activation convolution kernel for VcvConvArgs.io == 7 (bit 0 / 1: x / y + res are 16-bit
// tensors; bit 2 / 3: that x / y is fp16, else bf16); see conv_pk_io.hip.
#define VCV_IO_INST 7
#include "conv_pk_io_inst.h"

// version.hip -- build identification for libvcvits_hip.so
#include "common.h"
extern "C" const char* vcv_version(void) {
  return "vcvits_hip gfx950 fp32-mfma(32x32x2) conv family; built " __DATE__ " " __TIME__;
}

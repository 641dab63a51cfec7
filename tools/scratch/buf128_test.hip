// probe: raw buffer_load_dwordx4 -- per-dword range check at the end of the buffer, 4-byte-aligned addresses
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i4 __attribute__((ext_vector_type(4)));
template <int SH> __global__ void k(const float* src, float* out, int nrec_bytes, int base_off) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)(src + base_off), 0, nrec_bytes, 0x00020000);
  const int voff = ((int)threadIdx.x * 4 - SH) * 4;
  f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, 0, 0));
  for (int e = 0; e < 4; ++e) out[threadIdx.x * 4 + e] = v[e];
}
int main() {
  float *src, *out, h[1024], o[256];
  for (int i = 0; i < 1024; ++i) h[i] = 1000 + i;
  hipMalloc(&src, 4096); hipMalloc(&out, 1024);
  hipMemcpy(src, h, 4096, hipMemcpyHostToDevice);
  for (int sh = 0; sh < 3; ++sh) for (int base = 64; base <= 67; ++base) {
    const int SH = sh == 0 ? 8 : sh == 1 ? 6 : 7;   // base alignment 0, 4, 8, 12 bytes mod 16
    if (sh == 0) hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, src, out, 101 * 4, base);
    if (sh == 1) hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, src, out, 101 * 4, base);
    if (sh == 2) hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, src, out, 101 * 4, base);  // 101 floats in range
    hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) { int f = i - SH; float want = (f >= 0 && f < 101) ? 1000 + base + f : 0.f; if (o[i] != want) { if (bad < 4) printf("  base %d elem %d (f=%d): got %g want %g\n", base, i, f, o[i], want); ++bad; } }
    printf("shift %d base %d: %d mismatches\n", SH, base, bad);
  }
  return 0;
}

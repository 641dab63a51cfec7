// feasibility probe: LDS-DMA builtins on gfx950 (16-byte global_load_lds, 4-byte raw_buffer_load_lds with
// out-of-range lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* __restrict__ a, const float* __restrict__ x, int nx, float* out) {
  extern __shared__ float smem[];
  const int tid = threadIdx.x;
  // A: 256 threads x 16 B = 4 KB contiguous
  __builtin_amdgcn_global_load_lds(a + tid * 4, (__attribute__((address_space(3))) void*)(smem + (tid & ~63) * 4), 16, 0, 0);
  // X: buffer load to LDS, lanes beyond nx must read 0
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, nx * 4, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + 1024 + (tid & ~63)), 4, (tid - 8) * 4, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = tid; i < 1024 + 256; i += 256) out[i] = smem[i];
}

int main() {
  std::vector<float> ha(1024), hx(200);
  for (int i = 0; i < 1024; ++i) ha[i] = i;
  for (int i = 0; i < 200; ++i) hx[i] = 1000 + i;
  float *a, *x, *o;
  hipMalloc(&a, 4096); hipMalloc(&x, 800); hipMalloc(&o, (1024 + 256) * 4);
  hipMemcpy(a, ha.data(), 4096, hipMemcpyHostToDevice);
  hipMemcpy(x, hx.data(), 800, hipMemcpyHostToDevice);
  hipMemset(o, 0xff, (1024 + 256) * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), (1024 + 256) * 4, 0, a, x, 200, o);
  std::vector<float> ho(1024 + 256);
  hipMemcpy(ho.data(), o, ho.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (ho[i] != (float)i) { if (bad < 5) printf("A mismatch %d: %f\n", i, ho[i]); ++bad; }
  for (int t = 0; t < 256; ++t) {
    const int src = t - 8;
    const float want = (src >= 0 && src < 200) ? 1000.f + src : 0.f;
    if (ho[1024 + t] != want) { if (bad < 10) printf("X mismatch lane %d: got %f want %f\n", t, ho[1024 + t], want); ++bad; }
  }
  printf("bad=%d\n", bad);
  return bad != 0;
}

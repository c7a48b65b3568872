// Does gfx950 execute scalar memory atomics (s_atomic_add)? One wave per workgroup draws tickets from a counter with the scalar unit.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out, unsigned *ctr) {
  unsigned t;
  asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(ctr) : "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = t;
}
int main() {
  unsigned *d_out, *d_ctr, h[4096];
  hipMalloc(&d_out, sizeof h); hipMalloc(&d_ctr, 4); hipMemset(d_ctr, 0, 4);
  hipLaunchKernelGGL(k, dim3(4096), dim3(64), 0, 0, d_out, d_ctr);
  hipError_t e = hipDeviceSynchronize();
  printf("sync: %s\n", hipGetErrorString(e));
  unsigned c = 0; hipMemcpy(&c, d_ctr, 4, hipMemcpyDeviceToHost); hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
  long long sum = 0; unsigned mx = 0; for (unsigned v : h) { sum += v; if (v > mx) mx = v; }
  printf("counter %u (want 4096), tickets sum %lld (want %lld), max %u\n", c, sum, 4095LL * 4096 / 2, mx);
  return 0;
}

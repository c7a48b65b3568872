// copy_rate.hip -- which streaming-copy form reaches the on-box HBM ceiling (bench.py's hbm_copy_ceiling)?
// Build: hipcc --offload-arch=gfx950 -O3 -o copy_rate copy_rate.hip ; prints read+write GB/s of a 1 GiB copy per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

// A: one 16-B element per thread, grid = n / 256
__global__ __launch_bounds__(256) void k_a(uint4 *__restrict__ d, const uint4 *__restrict__ s, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) d[i] = s[i];
}
// B: grid-stride, U loads in flight per thread, each a full grid apart
template <int U> __global__ __launch_bounds__(256) void k_b(uint4 *__restrict__ d, const uint4 *__restrict__ s, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256; size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) { uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; u++) d[i + u * stride] = v[u]; }
  for (; i < n; i += stride) d[i] = s[i];
}
// C: each workgroup owns a contiguous chunk of U * 4 KiB; U loads in flight, 4 KiB apart
template <int U, bool NT> __global__ __launch_bounds__(256) void k_c(uint4 *__restrict__ d, const uint4 *__restrict__ s, size_t n) {
  for (size_t base = (size_t)blockIdx.x * 256 * U; base < n; base += (size_t)gridDim.x * 256 * U) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = base + u * 256 + threadIdx.x; if (i < n) { if (NT) { const u4v t = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(&s[i])); v[u] = make_uint4(t.x, t.y, t.z, t.w); } else v[u] = s[i]; } }
#pragma unroll
    for (int u = 0; u < U; u++) { const size_t i = base + u * 256 + threadIdx.x; if (i < n) { if (NT) { u4v t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, reinterpret_cast<u4v *>(&d[i])); } else d[i] = v[u]; } }
  }
}
typedef void (*kfn)(uint4 *, const uint4 *, size_t);
int main() {
  const size_t bytes = 1ull << 30, n = bytes / 16;
  uint4 *a, *b; CHK(hipMalloc(&a, bytes)); CHK(hipMalloc(&b, bytes)); CHK(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  struct { const char *name; kfn f; size_t per_block; int fixed_grid; } tab[] = {
    {"A one-per-thread", k_a, 256, 0},
    {"B grid-stride x4, 4096 wgs", k_b<4>, 0, 4096}, {"B grid-stride x4, 8192 wgs", k_b<4>, 0, 8192}, {"B grid-stride x8, 2048 wgs", k_b<8>, 0, 2048},
    {"B grid-stride x2, 16384 wgs", k_b<2>, 0, 16384},
    {"C chunk x4 (16 KiB/wg-iter), 2048 wgs", k_c<4, false>, 0, 2048}, {"C chunk x4, 4096 wgs", k_c<4, false>, 0, 4096}, {"C chunk x4, 8192 wgs", k_c<4, false>, 0, 8192},
    {"C chunk x8, 2048 wgs", k_c<8, false>, 0, 2048}, {"C chunk x8, 4096 wgs", k_c<8, false>, 0, 4096},
    {"C chunk x4 one-shot grid", k_c<4, false>, 1024, 0}, {"C chunk x8 one-shot grid", k_c<8, false>, 2048, 0},
    {"C chunk x4 nontemporal, 4096 wgs", k_c<4, true>, 0, 4096}, {"C chunk x8 nontemporal, 2048 wgs", k_c<8, true>, 0, 2048},
    {"C chunk x4 nontemporal one-shot", k_c<4, true>, 1024, 0},
  };
  for (auto &t : tab) {
    const unsigned grid = t.fixed_grid ? (unsigned)t.fixed_grid : (unsigned)((n + t.per_block - 1) / t.per_block);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(t.f, dim3(grid), dim3(256), 0, 0, b, a, n);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(t.f, dim3(grid), dim3(256), 0, 0, b, a, n);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-42s %8.1f GB/s (read+write)\n", t.name, 2.0 * bytes / (ms / 10 * 1e-3) / 1e9);
  }
  CHK(hipEventRecord(e0));
  for (int r = 0; r < 10; r++) CHK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-42s %8.1f GB/s (read+write)\n", "hipMemcpyAsync D2D", 2.0 * bytes / (ms / 10 * 1e-3) / 1e9);
  return 0;
}

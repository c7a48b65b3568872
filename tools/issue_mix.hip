// issue_mix.hip -- what does one VALU instruction of K1's mix cost at K1's occupancy, and what does an MFMA beside it cost?
// ("Why no MFMA")  Build: hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip
//   issue_mix [waves_per_simd ...]      (default: 2 3 4 8)
// All kernels: 256-thread workgroups, W of them per CU = W waves per SIMD, inline asm so that nothing is re-scheduled.
//   mix       the colour + row-DCT instruction mix of k_transform on 8 independent chains per lane
//             (cvt_f32_ubyte, 3 x fma, cvt_i32_f32, add, dot2_i32_i16, perm, ashr: 72 instructions per iteration)
//   mix_dep2 / mix_dep1   the same instructions as two / one dependent chain(s) per lane
//   fma cvt cvti dot2 perm add   single-opcode streams (8 chains): the per-opcode price at this occupancy
//   mix+mfma/N   the mix with one v_mfma_i32_16x16x64_i8 after every N = 8, 24, 72 VALU instructions
//   mfma      the MFMA alone, back to back on one accumulator
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define N_ITER 2048
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

#define I0(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define I1(n) "v_fma_f32 %" #n ", %" #n ", %[b], %[c]\n"
#define I4(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I5(n) "v_add_u32 %" #n ", %" #n ", %[b]\n"
#define I6(n) "v_dot2_i32_i16 %" #n ", %" #n ", %[b], %[c]\n"
#define I7(n) "v_perm_b32 %" #n ", %" #n ", %[b], %[c]\n"
#define I8(n) "v_ashrrev_i32 %" #n ", 11, %" #n "\n"
#define ROW(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define ROW2(I) I(0) I(1) I(0) I(1) I(0) I(1) I(0) I(1)
#define ROW1(I) I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0)
#define OUTS "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
#define INS [b] "v"(b), [c] "v"(c)
#define PRO uint32_t a[8]; for (int i = 0; i < 8; i++) a[i] = b + threadIdx.x * 7 + i;
#define EPI uint32_t r = 0; for (int i = 0; i < 8; i++) r += a[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = r;
#define MIXOF(R) R(I0) R(I1) R(I1) R(I1) R(I4) R(I5) R(I6) R(I7) R(I8)

#define PLAIN(NAME, BODY) __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c) { \
  PRO for (int it = 0; it < N_ITER; it++) asm volatile(BODY : OUTS : INS : "vcc"); EPI }
PLAIN(k_mix, MIXOF(ROW))
PLAIN(k_mix_dep2, MIXOF(ROW2))
PLAIN(k_mix_dep1, MIXOF(ROW1))
#define NINE(I) ROW(I) ROW(I) ROW(I) ROW(I) ROW(I) ROW(I) ROW(I) ROW(I) ROW(I)
PLAIN(k_fma, NINE(I1)) PLAIN(k_cvt, NINE(I0)) PLAIN(k_cvti, NINE(I4)) PLAIN(k_dot2, NINE(I6)) PLAIN(k_perm, NINE(I7)) PLAIN(k_add, NINE(I5))

#define MF "v_mfma_i32_16x16x64_i8 %[acc], %[ma], %[mb], %[acc]\n"
#define MKERNEL(NAME, BODY) __global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c) { \
  PRO v4i acc = {0, 0, 0, 0}, ma = {(int)b, 1, 2, 3}, mb = {(int)c, 4, 5, 6}; \
  for (int it = 0; it < N_ITER; it++) asm volatile(BODY : OUTS, [acc] "+v"(acc) : INS, [ma] "v"(ma), [mb] "v"(mb) : "vcc"); \
  asm volatile("s_nop 7\ns_nop 7\ns_nop 7" : "+v"(acc)); a[0] += acc.x + acc.y + acc.z + acc.w; EPI }
MKERNEL(k_mix_mfma8, ROW(I0) MF ROW(I1) MF ROW(I1) MF ROW(I1) MF ROW(I4) MF ROW(I5) MF ROW(I6) MF ROW(I7) MF ROW(I8) MF)
MKERNEL(k_mix_mfma24, ROW(I0) ROW(I1) ROW(I1) MF ROW(I1) ROW(I4) ROW(I5) MF ROW(I6) ROW(I7) ROW(I8) MF)
MKERNEL(k_mix_mfma72, MIXOF(ROW) MF)
MKERNEL(k_mfma, MF MF MF MF MF MF MF MF MF)

typedef void (*kfn)(uint32_t *, uint32_t, uint32_t);
int main(int argc, char **argv) {
  int cus = 256;
  CHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  uint32_t *d; CHK(hipMalloc(&d, (size_t)cus * 8 * 256 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  struct { const char *n; kfn f; int valu, mfma; } tab[] = {
    {"mix", k_mix, 72, 0}, {"mix_dep2", k_mix_dep2, 72, 0}, {"mix_dep1", k_mix_dep1, 72, 0}, {"fma", k_fma, 72, 0}, {"cvt_ubyte", k_cvt, 72, 0},
    {"cvt_i32", k_cvti, 72, 0}, {"dot2", k_dot2, 72, 0}, {"perm", k_perm, 72, 0}, {"add", k_add, 72, 0},
    {"mix+mfma/8", k_mix_mfma8, 72, 9}, {"mix+mfma/24", k_mix_mfma24, 72, 3}, {"mix+mfma/72", k_mix_mfma72, 72, 1}, {"mfma", k_mfma, 0, 9}};
  int ws_default[] = {2, 3, 4, 8};
  int nws = argc > 1 ? argc - 1 : 4;
  for (int wi = 0; wi < nws; wi++) {
    const int W = argc > 1 ? atoi(argv[1 + wi]) : ws_default[wi];
    const int blocks = cus * W;
    printf("---- %d waves per SIMD (%d workgroups of 256) ----\n", W, blocks);
    for (auto &t : tab) {
      hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
      CHK(hipDeviceSynchronize());
      float best = 1e9;
      for (int r = 0; r < 3; r++) {
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      // SIMD-cycles (at a nominal 2.4 GHz) that one SIMD spends per iteration of ONE wave's body, divided by the waves on it
      const double cyc_iter = best * 1e-3 * 2.4e9 / N_ITER / W;
      printf("%-12s %7.3f ms  %7.1f SIMD-cycles per wave-iteration", t.n, best, cyc_iter);
      if (t.valu) printf("  = %5.2f per VALU instruction%s", cyc_iter / t.valu, t.mfma ? " (MFMAs not counted)" : "");
      else printf("  = %5.2f per MFMA", cyc_iter / t.mfma);
      printf("\n");
    }
  }
  return 0;
}

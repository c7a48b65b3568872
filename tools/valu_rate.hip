// Microbenchmark: sustained issue cost (SIMD cycles per wave64 instruction) of the VALU ops the JPEG kernels use,
// on gfx950. Inline asm, 8 independent chains per lane, 8 waves per SIMD, so the number is pure issue throughput.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 1; } } while (0)

#define BODY8(INS) \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) \
    : "v"(b), "v"(c) : "vcc");

#define I_ADD(n)      "v_add_u32 %" #n ", %" #n ", %16\n"
#define I_AND(n)      "v_and_b32 %" #n ", %" #n ", %16\n"
#define I_LSHL(n)     "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define I_LSHLV(n)    "v_lshlrev_b32 %" #n ", %16, %" #n "\n"
#define I_ASHR(n)     "v_ashrrev_i32 %" #n ", 31, %" #n "\n"
#define I_LSHLOR(n)   "v_lshl_or_b32 %" #n ", %" #n ", 4, %16\n"
#define I_LSHLADD(n)  "v_lshl_add_u32 %" #n ", %" #n ", 2, %16\n"
#define I_ADD3(n)     "v_add3_u32 %" #n ", %" #n ", %16, %17\n"
#define I_OR3(n)      "v_or3_b32 %" #n ", %" #n ", %16, %17\n"
#define I_MAD24(n)    "v_mad_i32_i24 %" #n ", %" #n ", %16, %17\n"
#define I_MUL24(n)    "v_mul_i32_i24 %" #n ", %" #n ", %16\n"
#define I_MADU24(n)   "v_mad_u32_u24 %" #n ", %" #n ", %16, %17\n"
#define I_MULLO(n)    "v_mul_lo_u32 %" #n ", %" #n ", %16\n"
#define I_MULHI24(n)  "v_mul_hi_u32_u24 %" #n ", %" #n ", %16\n"
#define I_BFEU(n)     "v_bfe_u32 %" #n ", %" #n ", 3, 9\n"
#define I_BFEUV(n)    "v_bfe_u32 %" #n ", %" #n ", 0, %16\n"
#define I_BFEI(n)     "v_bfe_i32 %" #n ", %" #n ", 0, 16\n"
#define I_CNDMASK(n)  "v_cndmask_b32 %" #n ", %" #n ", %16, vcc\n"
#define I_CMP(n)      "v_cmp_lt_i32 vcc, %" #n ", %16\n"
#define I_CVTFI(n)    "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define I_CVTIF(n)    "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_FMA(n)      "v_fma_f32 %" #n ", %" #n ", %16, %17\n"
#define I_FREXP(n)    "v_frexp_exp_i32_f32 %" #n ", %" #n "\n"
#define I_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %16, %17\n"
#define I_PERM(n)     "v_perm_b32 %" #n ", %" #n ", %16, %17\n"
#define I_MAX(n)      "v_max_i32 %" #n ", %" #n ", %16\n"
#define I_FFBH(n)     "v_ffbh_u32 %" #n ", %" #n "\n"
#define I_BCNT(n)     "v_bcnt_u32_b32 %" #n ", %" #n ", %16\n"
#define I_BFI(n)      "v_bfi_b32 %" #n ", %" #n ", %16, %17\n"
#define I_PKADD(n)    "v_pk_add_i16 %" #n ", %" #n ", %16\n"
#define I_PKMUL(n)    "v_pk_mul_lo_u16 %" #n ", %" #n ", %16\n"
#define I_PKMAD(n)    "v_pk_mad_i16 %" #n ", %" #n ", %16, %17\n"
#define I_PKASHR(n)   "v_pk_ashrrev_i16 %" #n ", 3, %" #n "\n"
#define I_PKMAX(n)    "v_pk_max_i16 %" #n ", %" #n ", %16\n"
#define I_MADI16(n)   "v_mad_i16 %" #n ", %" #n ", %16, %17\n"
#define I_SDWA(n)     "v_add_u32_sdwa %" #n ", %" #n ", %16 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define I_SHL64(n)    "v_lshlrev_b64 %" #n ", %16, %" #n "\n"
#define I_PKFMA(n)    "v_pk_fma_f32 %" #n ", %" #n ", %" #n ", %" #n "\n"
#define I_PKMULF(n)   "v_pk_mul_f32 %" #n ", %" #n ", %" #n "\n"
#define I_PKADDF(n)   "v_pk_add_f32 %" #n ", %" #n ", %" #n "\n"
#define I_MAXU(n)     "v_max_u32 %" #n ", %" #n ", %16\n"
#define I_MINU(n)     "v_min_u32 %" #n ", %" #n ", %16\n"
#define I_CVTUB(n)    "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define I_ADDF(n)     "v_add_f32 %" #n ", %" #n ", %16\n"
#define I_XOR(n)      "v_xor_b32 %" #n ", %" #n ", %16\n"
#define I_SUB(n)      "v_sub_u32 %" #n ", %" #n ", %16\n"
#define I_DOT4(n)     "v_dot4_i32_i8 %" #n ", %" #n ", %16, %17\n"
#define I_DOT2(n)     "v_dot2_i32_i16 %" #n ", %" #n ", %16, %17\n"
#define I_CVTPK(n)    "v_cvt_pk_i16_i32 %" #n ", %" #n ", %16\n"
#define I_MOV(n)      "v_mov_b32 %" #n ", %16\n"
#define I_SAD(n)      "v_sad_u8 %" #n ", %" #n ", %16, %17\n"
#define I_MED3(n)     "v_med3_i32 %" #n ", %" #n ", %16, %17\n"

#define KERNEL(NAME, INS, WIDE) \
__global__ __launch_bounds__(256) void NAME(uint32_t *out, uint32_t b, uint32_t c) { \
  uint32_t a[8]; unsigned long long w[8]; \
  for (int i = 0; i < 8; i++) { a[i] = b + threadIdx.x * 7 + i; w[i] = a[i]; } \
  for (int it = 0; it < N_ITER; it++) { if (WIDE) { asm volatile(INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) : "v"(b), "v"(c) : "vcc"); } else { BODY8(INS) } } \
  uint32_t r = 0; for (int i = 0; i < 8; i++) r += a[i] + (uint32_t)w[i]; \
  out[blockIdx.x * blockDim.x + threadIdx.x] = r; }

#define LIST(X) X(k_add, I_ADD, 0) X(k_and, I_AND, 0) X(k_lshl, I_LSHL, 0) X(k_lshlv, I_LSHLV, 0) X(k_ashr, I_ASHR, 0) X(k_lshlor, I_LSHLOR, 0) \
  X(k_lshladd, I_LSHLADD, 0) X(k_add3, I_ADD3, 0) X(k_or3, I_OR3, 0) X(k_mad24, I_MAD24, 0) X(k_mul24, I_MUL24, 0) X(k_madu24, I_MADU24, 0) \
  X(k_mullo, I_MULLO, 0) X(k_mulhi24, I_MULHI24, 0) X(k_bfeu, I_BFEU, 0) X(k_bfeuv, I_BFEUV, 0) X(k_bfei, I_BFEI, 0) X(k_cndmask, I_CNDMASK, 0) X(k_cmp, I_CMP, 0) \
  X(k_cvtfi, I_CVTFI, 0) X(k_cvtif, I_CVTIF, 0) X(k_fma, I_FMA, 0) X(k_frexp, I_FREXP, 0) X(k_alignbit, I_ALIGNBIT, 0) X(k_perm, I_PERM, 0) \
  X(k_max, I_MAX, 0) X(k_ffbh, I_FFBH, 0) X(k_bcnt, I_BCNT, 0) X(k_bfi, I_BFI, 0) X(k_pkadd, I_PKADD, 0) X(k_pkmul, I_PKMUL, 0) X(k_pkmad, I_PKMAD, 0) \
  X(k_pkashr, I_PKASHR, 0) X(k_pkmax, I_PKMAX, 0) X(k_madi16, I_MADI16, 0) X(k_sdwa, I_SDWA, 0) X(k_shl64, I_SHL64, 1) X(k_dot4, I_DOT4, 0) X(k_dot2, I_DOT2, 0) \
  X(k_pkfma, I_PKFMA, 1) X(k_pkmulf, I_PKMULF, 1) X(k_pkaddf, I_PKADDF, 1) X(k_maxu, I_MAXU, 0) X(k_minu, I_MINU, 0) X(k_cvtub, I_CVTUB, 0) X(k_addf, I_ADDF, 0) X(k_xor, I_XOR, 0) X(k_sub, I_SUB, 0) X(k_cvtpk, I_CVTPK, 0) X(k_mov, I_MOV, 0) X(k_sad, I_SAD, 0) X(k_med3, I_MED3, 0)

LIST(KERNEL)

typedef void (*kfn)(uint32_t *, uint32_t, uint32_t);
int main() {
  const int blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
  uint32_t *d; CHK(hipMalloc(&d, blocks * 256 * 4));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
#define ENTRY(NAME, INS, WIDE) {#NAME, NAME},
  struct { const char *n; kfn f; } tab[] = { LIST(ENTRY) };
  // clock estimate: v_fma_f32 is documented as 2 cycles/wave64 at >= 2 waves per SIMD
  for (auto &t : tab) {
    hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u);
    CHK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
      CHK(hipEventRecord(e0)); hipLaunchKernelGGL(t.f, dim3(blocks), dim3(256), 0, 0, d, 3u, 5u); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double insts = (double)blocks * 4 * N_ITER * 8, cyc = best * 1e-3 * 2.4e9 * 1024;
    printf("%-10s %7.3f ms  %5.2f SIMD-cycles/instr @2.4GHz\n", t.n + 2, best, cyc / insts);
  }
  return 0;
}

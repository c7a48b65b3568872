// mij_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4): the JPEG encode hot path.
//
// Replaces what the reference gets from nvjpegEncodeImage (reference ImageCompressorImpl.cu:280):
//   K1 k_transform     BGR/RGB -> YCbCr, chroma downsample, level shift, 8x8 FDCT, quantise, zig-zag   (SURVEY 8a A3+A4)
//   K2 k_histogram     DC-difference / AC run-length symbol statistics                                 (A5+A6)
//   K3 k_build_tables  optimal (or Annex K) Huffman tables, encoder LUT, JFIF header                   (A6+A7)
//   K4 k_encode        Huffman coding + bit packing, one restart interval per wavefront                (A5+A7)
//   K5 k_scan          exclusive scan of interval sizes                                                (A7)
//   K6 k_compact       FF00 byte stuffing + compaction + RSTn / EOI markers                            (A7)
// No MFMA anywhere: this path is integer/byte work bounded by HBM and by VALU issue, not a contraction.
// Wavefront = 64 lanes throughout.
#include "mij_internal.h"

namespace mij {

// ---------------------------------------------------------------------------------------------------------------
// Small device helpers
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ int mad24(int a, int b, int c) { return __mul24(a, b) + c; }

__device__ __constant__ uint8_t c_zigzag[64] = {
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
    28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
    54, 47, 55, 62, 63};

// compile-time zig-zag (for fully unrolled register indexing)
__host__ __device__ constexpr int zz(int k) {
  constexpr int t[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                         41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                         30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
  return t[k];
}

// 8-point forward DCT, "accurate integer" LLM factorisation with 13-bit constants (T.81 compatible; the same
// arithmetic as oracle/jpeg_oracle.c fdct_islow, which is pinned against libjpeg-turbo / IJG).
// PASS 1 (rows): outputs scaled by 2^2.  PASS 2 (columns): removes that scaling, leaving the overall factor 8.
// All products fit 24-bit x 24-bit -> 32-bit, so the full-rate v_mul_i32_i24 / v_mad_i32_i24 are used.
template <int PASS>
__device__ __forceinline__ void dct8(int &d0, int &d1, int &d2, int &d3, int &d4, int &d5, int &d6, int &d7) {
  constexpr int SH = PASS == 1 ? 11 : 15;
  constexpr int RND = 1 << (SH - 1);
  int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
  int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  if (PASS == 1) {
    d0 = (t10 + t11) << 2;
    d4 = (t10 - t11) << 2;
  } else {
    d0 = (t10 + t11 + 2) >> 2;
    d4 = (t10 - t11 + 2) >> 2;
  }
  int z1 = mad24(t12 + t13, 4433, RND);
  d2 = mad24(t13, 6270, z1) >> SH;
  d6 = mad24(t12, -15137, z1) >> SH;
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  int z5 = mad24(z3 + z4, 9633, RND);
  z3 = mad24(z3, -16069, z5);
  z4 = mad24(z4, -3196, z5);
  z1 = mul24(z1, -7373);
  z2 = mul24(z2, -20995);
  d7 = (mad24(t4, 2446, z1) + z3) >> SH;
  d5 = (mad24(t5, 16819, z2) + z4) >> SH;
  d3 = (mad24(t6, 25172, z2) + z3) >> SH;
  d1 = (mad24(t7, 12299, z1) + z4) >> SH;
}

// Quantise: sign(t) * floor((|t| + 4q) / 8q), evaluated exactly in fp32 as trunc(t * r +- b) with r = 1/(8q),
// b = (4q + 0.5) r  (exactness over |t| < 2^22 is argued in DESIGN.md and brute-forced in tests/test_quant_exact.py).
__device__ __forceinline__ int quant1(int t, float r, float b) {
  float tf = (float)t;
  float bs = __builtin_copysignf(b, tf);
  return (int)__builtin_fmaf(tf, r, bs);
}

// Column pass + quantise + zig-zag + pack of a block whose row pass is already done; stores 128 B.
__device__ __forceinline__ void finish_block(int (&d)[64], const Quant *__restrict__ qt, int tbl,
                                             int16_t *__restrict__ dst) {
#pragma unroll
  for (int c = 0; c < 8; c++)
    dct8<2>(d[c], d[8 + c], d[16 + c], d[24 + c], d[32 + c], d[40 + c], d[48 + c], d[56 + c]);
  uint32_t w[32];
#pragma unroll
  for (int k = 0; k < 64; k += 2) {
    constexpr int dummy = 0;
    (void)dummy;
    int n0 = zz(k), n1 = zz(k + 1);
    int q0 = quant1(d[n0], qt->recip[tbl][n0], qt->bias[tbl][n0]);
    int q1 = quant1(d[n1], qt->recip[tbl][n1], qt->bias[tbl][n1]);
    w[k >> 1] = ((uint32_t)q0 & 0xFFFFu) | ((uint32_t)q1 << 16);
  }
  uint4 *o = reinterpret_cast<uint4 *>(dst);
#pragma unroll
  for (int i = 0; i < 8; i++) o[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

// ---------------------------------------------------------------------------------------------------------------
// K1: transform. One workgroup = 256 threads = 256 luma blocks = 256/(HS*VS) MCUs.
// Phase 1: thread t owns one 8x8 pixel region: loads it, converts colour, runs the luma block through
//          FDCT+quantise, and reduces its region's Cb/Cr to the subsampled resolution, which it drops into LDS.
// Phase 2: the first 2*MCUs threads each pick one finished 8x8 chroma block out of LDS and transform it.
// ---------------------------------------------------------------------------------------------------------------
template <int HS, int VS>
struct TCfg {
  static constexpr int NL = HS * VS, MPT = 256 / NL, BPM = NL + 2, CW = 8 / HS, CH = 8 / VS;
};

template <bool INTERLEAVED>
__device__ __forceinline__ void load_row8(const TransformArgs &a, int ysrc, int x0, int W, bool edge, int amode,
                                          int (&A)[8], int (&G)[8], int (&C)[8]) {
  if (INTERLEAVED) {
    const uint8_t *row = a.src + (size_t)ysrc * a.pitch;
    if (!edge && amode >= 4) {
      uint32_t w[6];
      const uint8_t *p = row + (size_t)x0 * 3;
      if (amode == 8) {
        const uint2 *q = reinterpret_cast<const uint2 *>(p);
        uint2 u0 = q[0], u1 = q[1], u2 = q[2];
        w[0] = u0.x; w[1] = u0.y; w[2] = u1.x; w[3] = u1.y; w[4] = u2.x; w[5] = u2.y;
      } else {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p);
#pragma unroll
        for (int i = 0; i < 6; i++) w[i] = q[i];
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int b0 = 3 * i, b1 = 3 * i + 1, b2 = 3 * i + 2;
        A[i] = (w[b0 >> 2] >> ((b0 & 3) * 8)) & 255;
        G[i] = (w[b1 >> 2] >> ((b1 & 3) * 8)) & 255;
        C[i] = (w[b2 >> 2] >> ((b2 & 3) * 8)) & 255;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int x = min(x0 + i, W - 1);
        const uint8_t *p = row + (size_t)x * 3;
        A[i] = p[0]; G[i] = p[1]; C[i] = p[2];
      }
    }
  } else {
    const uint8_t *r0 = a.src + (size_t)ysrc * a.pitch, *r1 = r0 + a.plane_stride, *r2 = r1 + a.plane_stride;
    if (!edge && amode >= 4) {
      uint32_t w[3][2];
      if (amode == 8) {
        uint2 u0 = *reinterpret_cast<const uint2 *>(r0 + x0), u1 = *reinterpret_cast<const uint2 *>(r1 + x0),
              u2 = *reinterpret_cast<const uint2 *>(r2 + x0);
        w[0][0] = u0.x; w[0][1] = u0.y; w[1][0] = u1.x; w[1][1] = u1.y; w[2][0] = u2.x; w[2][1] = u2.y;
      } else {
        const uint32_t *q0 = reinterpret_cast<const uint32_t *>(r0 + x0), *q1 = reinterpret_cast<const uint32_t *>(r1 + x0),
                       *q2 = reinterpret_cast<const uint32_t *>(r2 + x0);
        w[0][0] = q0[0]; w[0][1] = q0[1]; w[1][0] = q1[0]; w[1][1] = q1[1]; w[2][0] = q2[0]; w[2][1] = q2[1];
      }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        A[i] = (w[0][i >> 2] >> ((i & 3) * 8)) & 255;
        G[i] = (w[1][i >> 2] >> ((i & 3) * 8)) & 255;
        C[i] = (w[2][i >> 2] >> ((i & 3) * 8)) & 255;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        int x = min(x0 + i, W - 1);
        A[i] = r0[x]; G[i] = r1[x]; C[i] = r2[x];
      }
    }
  }
}

template <int HS, int VS, bool INTERLEAVED>
__global__ __launch_bounds__(256) void k_transform(const Geom g, const TransformArgs a, const int amode) {
  using Cfg = TCfg<HS, VS>;
  constexpr int NL = Cfg::NL, MPT = Cfg::MPT, BPM = Cfg::BPM, CW = Cfg::CW, CH = Cfg::CH;
  __shared__ uint32_t s_chroma[2][MPT][16];  // [Cb|Cr][mcu in tile][8 rows x 8 bytes]

  const int t = threadIdx.x;
  const long long tile_first = (long long)blockIdx.x * MPT;  // MCU index relative to the strip
  // ---------------- phase 1 ----------------
  {
    const int m = t / NL, s = t % NL, sx = s % HS, sy = s / HS;
    const long long ml = tile_first + m;
    if (ml < g.mcu_count) {
      const long long gm = g.mcu_first + ml;
      const int mx = (int)(gm % g.mcux), my = (int)(gm / g.mcux);
      const int bx = mx * HS + sx, by = my * VS + sy;
      const int x0 = bx * 8, y0 = by * 8;
      const bool edge = (x0 + 8 > g.W) || (y0 + 8 > g.H);
      const bool real = (bx < g.wib0) && (by < g.hib0);
      int d[64];
      uint8_t *cbp = reinterpret_cast<uint8_t *>(&s_chroma[0][m][0]);
      uint8_t *crp = reinterpret_cast<uint8_t *>(&s_chroma[1][m][0]);
      int scb[CW], scr[CW];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        int A[8], G[8], C[8];
        const int yl = min(y0 + j, g.H - 1);
        load_row8<INTERLEAVED>(a, yl - g.y_origin, x0, g.W, edge, amode, A, G, C);
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int y = mad24(A[i], a.kA[0], mad24(G[i], 38470, mad24(C[i], a.kC[0], 32768))) >> 16;
          d[j * 8 + i] = y - 128;
        }
        dct8<1>(d[j * 8], d[j * 8 + 1], d[j * 8 + 2], d[j * 8 + 3], d[j * 8 + 4], d[j * 8 + 5], d[j * 8 + 6], d[j * 8 + 7]);
        if (edge) {
          // Bottom edge: chroma rows past the last really-downsampled row replicate that OUTPUT row, which is not the
          // same as downsampling replicated input rows (see oracle build_component_plane). Re-fetch when they differ.
          const int oy = min((y0 + j) / VS, g.crows - 1);
          const int yc = min(oy * VS + (j % VS), g.H - 1);
          if (yc != yl) load_row8<INTERLEAVED>(a, yc - g.y_origin, x0, g.W, edge, amode, A, G, C);
        }
        if (j % VS == 0) {
#pragma unroll
          for (int ii = 0; ii < CW; ii++) { scb[ii] = 0; scr[ii] = 0; }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
          int cb = mad24(A[i], a.kA[1], mad24(G[i], -21709, mad24(C[i], a.kC[1], 8421375))) >> 16;
          int cr = mad24(A[i], a.kA[2], mad24(G[i], -27439, mad24(C[i], a.kC[2], 8421375))) >> 16;
          scb[i / HS] += cb;
          scr[i / HS] += cr;
        }
        if (j % VS == VS - 1) {  // row group complete: finalise the subsampled chroma row and drop it into LDS
          const int jj = j / VS;
          uint32_t pcb[2] = {0, 0}, pcr[2] = {0, 0};
#pragma unroll
          for (int ii = 0; ii < CW; ii++) {
            int vb, vr;
            if (HS == 1 && VS == 1) { vb = scb[ii]; vr = scr[ii]; }
            else if (HS == 2 && VS == 1) { vb = (scb[ii] + (ii & 1)) >> 1; vr = (scr[ii] + (ii & 1)) >> 1; }
            else if (HS == 2 && VS == 2) { vb = (scb[ii] + 1 + (ii & 1)) >> 2; vr = (scr[ii] + 1 + (ii & 1)) >> 2; }
            else { constexpr int N = HS * VS; vb = (scb[ii] + N / 2) / N; vr = (scr[ii] + N / 2) / N; }
            pcb[ii >> 2] |= (uint32_t)vb << ((ii & 3) * 8);
            pcr[ii >> 2] |= (uint32_t)vr << ((ii & 3) * 8);
          }
          const int row = sy * CH + jj, col = sx * CW;
          if (CW == 8) {
            *reinterpret_cast<uint2 *>(cbp + row * 8) = make_uint2(pcb[0], pcb[1]);
            *reinterpret_cast<uint2 *>(crp + row * 8) = make_uint2(pcr[0], pcr[1]);
          } else if (CW == 4) {
            *reinterpret_cast<uint32_t *>(cbp + row * 8 + col) = pcb[0];
            *reinterpret_cast<uint32_t *>(crp + row * 8 + col) = pcr[0];
          } else {
            *reinterpret_cast<uint16_t *>(cbp + row * 8 + col) = (uint16_t)pcb[0];
            *reinterpret_cast<uint16_t *>(crp + row * 8 + col) = (uint16_t)pcr[0];
          }
        }
      }
      int16_t *dst = a.coef + ((size_t)ml * BPM + s) * 64;
      if (real) {
        finish_block(d, a.qt, 0, dst);
      } else {  // dummy block: AC = 0; DC is patched by k_fix_dummy_dc
        uint4 *o = reinterpret_cast<uint4 *>(dst);
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = make_uint4(0, 0, 0, 0);
      }
    }
  }
  __syncthreads();
  // ---------------- phase 2: chroma blocks ----------------
  for (int u = t; u < 2 * MPT; u += 256) {
    const int comp = u / MPT, m = u % MPT;
    const long long ml = tile_first + m;
    if (ml >= g.mcu_count) continue;
    const uint4 *sp = reinterpret_cast<const uint4 *>(&s_chroma[comp][m][0]);
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 4; i++) { uint4 v = sp[i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
    int d[64];
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
      for (int i = 0; i < 8; i++) d[j * 8 + i] = (int)((w[j * 2 + (i >> 2)] >> ((i & 3) * 8)) & 255) - 128;
      dct8<1>(d[j * 8], d[j * 8 + 1], d[j * 8 + 2], d[j * 8 + 3], d[j * 8 + 4], d[j * 8 + 5], d[j * 8 + 6], d[j * 8 + 7]);
    }
    finish_block(d, a.qt, 1, a.coef + ((size_t)ml * BPM + NL + comp) * 64);
  }
}

// Dummy luma blocks (image size not a multiple of the MCU size): DC := DC of the block the entropy coder will have
// seen just before in the same component, so the coded difference is 0 (same rule as the oracle / libjpeg).
// One thread per MCU of the last MCU column / last MCU row of the strip.
__global__ void k_fix_dummy_dc(const Geom g, int16_t *coef) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.mcu_count) return;
  const long long gm = g.mcu_first + i;
  const int mx = (int)(gm % g.mcux), my = (int)(gm / g.mcux);
  if (mx != g.mcux - 1 && my != g.mcuy - 1) return;
  int16_t *mcu = coef + (size_t)i * g.bpm * 64;
  for (int yi = 0; yi < g.vs; yi++)
    for (int xi = 0; xi < g.hs; xi++) {
      const int bx = mx * g.hs + xi, by = my * g.vs + yi;
      if (bx < g.wib0 && by < g.hib0) continue;
      int16_t *o = mcu + (yi * g.hs + xi) * 64;
      o[0] = (by < g.hib0) ? o[-64] : mcu[(yi * g.hs - 1) * 64];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Shared front end of K2/K4: a single-wave workgroup pulls 64 consecutive blocks (8 KiB, fully coalesced 16-B loads)
// through LDS so that lane L ends up holding block L's 64 coefficients in 32 packed registers.
// LDS row stride 144 B keeps both the 16-B writes and the 16-B reads conflict free.
// ---------------------------------------------------------------------------------------------------------------
constexpr int STAGE_STRIDE = 144;
constexpr int STAGE_BYTES = 64 * STAGE_STRIDE;

__device__ __forceinline__ void load_batch(const int16_t *__restrict__ base, int nvalid, uint8_t *lds, int lane,
                                           uint32_t (&c)[32]) {
  const uint4 *src = reinterpret_cast<const uint4 *>(base);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    const int p = i * 64 + lane, blk = p >> 3, j = p & 7;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (blk < nvalid) v = src[p];
    *reinterpret_cast<uint4 *>(lds + blk * STAGE_STRIDE + j * 16) = v;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 8; j++) {
    uint4 v = *reinterpret_cast<const uint4 *>(lds + lane * STAGE_STRIDE + j * 16);
    c[4 * j] = v.x; c[4 * j + 1] = v.y; c[4 * j + 2] = v.z; c[4 * j + 3] = v.w;
  }
  __syncthreads();
}

__device__ __forceinline__ int coef_at(const uint32_t (&c)[32], int k) {  // k compile-time after unrolling
  return (k & 1) ? ((int)c[k >> 1] >> 16) : (int)(int16_t)(c[k >> 1] & 0xFFFF);
}
__device__ __forceinline__ int nbits_of(int v) { return 32 - __clz(v < 0 ? -v : v); }

// DC predictor of block `lb` (index relative to the strip) inside the restart interval that starts at block seg_first.
__device__ __forceinline__ int dc_pred(const int16_t *__restrict__ coef, long long lb, long long seg_first, int bpm,
                                       int nl, int &comp_is_chroma) {
  const int bim = (int)(lb % bpm);
  comp_is_chroma = bim >= nl;
  const long long prev = comp_is_chroma ? lb - bpm : (bim > 0 ? lb - 1 : lb - bpm + nl - 1);
  return prev >= seg_first ? (int)coef[prev * 64] : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// K2: symbol statistics. One wave per restart interval, lane per block.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_histogram(const Geom g, const int16_t *__restrict__ coef,
                                                  uint32_t *__restrict__ hist) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[STAGE_BYTES];
  __shared__ uint32_t s_h[4 * 257];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4 * 257; i += 64) s_h[i] = 0;
  __syncthreads();
  const long long seg = blockIdx.x;
  const long long seg_first = seg * g.ri * g.bpm;
  const long long seg_end = min(seg_first + (long long)g.ri * g.bpm, g.mcu_count * g.bpm);
  for (long long b0 = seg_first; b0 < seg_end; b0 += 64) {
    const int nvalid = (int)min((long long)64, seg_end - b0);
    uint32_t c[32];
    load_batch(coef + b0 * 64, nvalid, s_stage, lane, c);
    if (lane < nvalid) {
      int chroma;
      const int pred = dc_pred(coef, b0 + lane, seg_first, g.bpm, g.nl, chroma);
      uint32_t *hdc = s_h + (chroma ? 2 : 0) * 257, *hac = s_h + (chroma ? 3 : 1) * 257;
      atomicAdd(&hdc[nbits_of(coef_at(c, 0) - pred)], 1u);
      int r = 0;
#pragma unroll
      for (int k = 1; k < 64; k++) {
        const int v = coef_at(c, k);
        if (v == 0) { r++; }
        else {
          if (r > 15) { atomicAdd(&hac[0xF0], (uint32_t)(r >> 4)); r &= 15; }
          atomicAdd(&hac[(r << 4) + nbits_of(v)], 1u);
          r = 0;
        }
      }
      if (r > 0) atomicAdd(&hac[0], 1u);
    }
  }
  __syncthreads();
  for (int i = lane; i < 4 * 257; i += 64) { const uint32_t v = s_h[i]; if (v) atomicAdd(&hist[i], v); }
}

// ---------------------------------------------------------------------------------------------------------------
// K3: Huffman tables + header. One workgroup of 4 waves; wave w builds table w
// (0 DC luma, 1 AC luma, 2 DC chroma, 3 AC chroma) with T.81 K.2's procedure, the argmin steps done wave-wide.
// Tie rule (pins byte-identity with libjpeg-turbo): among equal frequencies the larger symbol value is taken.
// ---------------------------------------------------------------------------------------------------------------
__device__ __constant__ uint8_t c_std_bits[4][17] = {
    {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d},
    {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0},
    {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}};
__device__ __constant__ uint8_t c_std_ac_l[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
__device__ __constant__ uint8_t c_std_ac_c[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

struct MinKey { uint32_t f; int i; };  // i < 0 : none
__device__ __forceinline__ bool better(uint32_t fa, int ia, uint32_t fb, int ib) {
  // is (fa, ia) a better "smallest" candidate than (fb, ib)?  none loses; smaller f wins; ties -> larger index
  if (ia < 0) return false;
  if (ib < 0) return true;
  return fa < fb || (fa == fb && ia > ib);
}
__device__ __forceinline__ MinKey wave_min(MinKey k) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const uint32_t of = __shfl_xor(k.f, off, 64);
    const int oi = __shfl_xor(k.i, off, 64);
    if (better(of, oi, k.f, k.i)) { k.f = of; k.i = oi; }
  }
  return k;
}

__global__ __launch_bounds__(256) void k_build_tables(const Geom g, const uint32_t *__restrict__ hist, const int optimize,
                                                      const Quant *__restrict__ qt, DeviceTables *__restrict__ tab,
                                                      uint8_t *__restrict__ out, DeviceResult *__restrict__ res) {
  __shared__ int s_cs[4][257];       // code size per symbol
  __shared__ int s_bits[4][33];
  __shared__ uint8_t s_vals[4][256];
  __shared__ int s_nvals[4];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_ac = w & 1;

  if (optimize) {
    // symbols owned by this lane: lane + 64 j, j = 0..4 (index 256 = reserved pseudo-symbol, lane 0 only)
    uint32_t f[5]; int cs[5], tr[5];
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const int i = lane + 64 * j;
      f[j] = i < 256 ? hist[w * 257 + i] : (i == 256 ? 1u : 0u);
      cs[j] = 0; tr[j] = i;
    }
    for (int iter = 0; iter < 257; iter++) {
      MinKey k1 = {0, -1};
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const int i = lane + 64 * j;
        if (i <= 256 && f[j] && better(f[j], i, k1.f, k1.i)) { k1.f = f[j]; k1.i = i; }
      }
      k1 = wave_min(k1);
      MinKey k2 = {0, -1};
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const int i = lane + 64 * j;
        if (i <= 256 && f[j] && i != k1.i && better(f[j], i, k2.f, k2.i)) { k2.f = f[j]; k2.i = i; }
      }
      k2 = wave_min(k2);
      if (k2.i < 0) break;
#pragma unroll
      for (int j = 0; j < 5; j++) {
        const int i = lane + 64 * j;
        if (i == k1.i) f[j] = k1.f + k2.f;
        if (i == k2.i) f[j] = 0;
        if (i <= 256 && (tr[j] == k1.i || tr[j] == k2.i)) { cs[j]++; tr[j] = k1.i; }
      }
    }
#pragma unroll
    for (int j = 0; j < 5; j++) { const int i = lane + 64 * j; if (i <= 256) s_cs[w][i] = cs[j]; }
    if (lane < 33) s_bits[w][lane] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 5; j++) { const int i = lane + 64 * j; if (i <= 256 && cs[j] > 0) atomicAdd(&s_bits[w][min(cs[j], 32)], 1); }
    __syncthreads();
    if (lane == 0) {  // Figure K.3: limit code lengths to 16, then drop the reserved code point
      int *bits = s_bits[w];
      int i;
      for (i = 32; i > 16; i--)
        while (bits[i] > 0) {
          int j = i - 2;
          while (bits[j] == 0) j--;
          bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
        }
      while (bits[i] == 0) i--;
      bits[i]--;
    }
    __syncthreads();
    // symbols sorted by (code size, symbol value): rank by counting
    int total = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int i = lane + 64 * j;
      const int my = cs[j];
      if (my > 0) {
        int rank = 0;
        for (int o = 0; o < 256; o++) {
          const int oc = s_cs[w][o];
          rank += (oc > 0 && (oc < my || (oc == my && o < i))) ? 1 : 0;
        }
        s_vals[w][rank] = (uint8_t)i;
      }
      total += my > 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) total += __shfl_xor(total, off, 64);
    if (lane == 0) s_nvals[w] = total;
  } else {
    if (lane < 17) s_bits[w][lane] = c_std_bits[w][lane];
    if (is_ac) {
      const uint8_t *v = (w == 1) ? c_std_ac_l : c_std_ac_c;
      for (int i = lane; i < 162; i += 64) s_vals[w][i] = v[i];
      if (lane == 0) s_nvals[w] = 162;
    } else {
      if (lane < 12) s_vals[w][lane] = (uint8_t)lane;
      if (lane == 0) s_nvals[w] = 12;
    }
  }
  __syncthreads();
  // canonical codes (T.81 Annex C) -> DHT payload + encoder LUT
  {
    const int lut_base = w == 0 ? LUT_DC_L : w == 1 ? LUT_AC_L : w == 2 ? LUT_DC_C : LUT_AC_C;
    const int lut_n = is_ac ? 256 : 16;
    for (int i = lane; i < lut_n; i += 64) tab->lut[lut_base + i] = 0;
    if (lane < 17) tab->bits[w][lane] = lane == 0 ? 0 : (uint8_t)s_bits[w][lane];
    const int nv = s_nvals[w];
    if (lane == 0) tab->nvals[w] = (uint32_t)nv;
    __syncthreads();
    for (int p = lane; p < 256; p += 64) {
      uint8_t sym = p < nv ? s_vals[w][p] : 0;
      tab->vals[w][p] = sym;
      if (p < nv) {
        int cum = 0, code = 0, l = 1;
        for (; l <= 16; l++) {
          const int b = s_bits[w][l];
          if (p < cum + b) { code += p - cum; break; }
          cum += b;
          code = (code + b) << 1;
        }
        if (l <= 16 && (int)sym < lut_n) tab->lut[lut_base + sym] = ((uint32_t)code << 5) | (uint32_t)l;
      }
    }
  }
  __syncthreads();
  // JFIF header, right-aligned in the first HDR_AREA bytes of `out` (marker order: SOI APP0 DQT DQT SOF0 DHT x4 DRI SOS)
  if (threadIdx.x == 0) {
    int len = 2 + 18 + 2 * 69 + 19 + 6 + 14;
    for (int t = 0; t < 4; t++) len += 5 + 16 + s_nvals[t];
    uint8_t *p = out + (HDR_AREA - len);
    auto put = [&](int b) { *p++ = (uint8_t)b; };
    auto put16 = [&](int v) { put(v >> 8); put(v & 255); };
    put16(0xFFD8);
    put16(0xFFE0); put16(16); put('J'); put('F'); put('I'); put('F'); put(0); put(1); put(1); put(0); put16(1); put16(1); put(0); put(0);
    for (int t = 0; t < 2; t++) {
      put16(0xFFDB); put16(67); put(t);
      for (int k = 0; k < 64; k++) put(qt->q[t][c_zigzag[k]]);
    }
    put16(0xFFC0); put16(17); put(8); put16(g.H); put16(g.W); put(3);
    put(1); put((g.hs << 4) | g.vs); put(0); put(2); put(0x11); put(1); put(3); put(0x11); put(1);
    const int tcth[4] = {0x00, 0x10, 0x01, 0x11};
    for (int t = 0; t < 4; t++) {
      put16(0xFFC4); put16(3 + 16 + s_nvals[t]); put(tcth[t]);
      for (int l = 1; l <= 16; l++) put(s_bits[t][l]);
      for (int i = 0; i < s_nvals[t]; i++) put(s_vals[t][i]);
    }
    put16(0xFFDD); put16(4); put16(g.ri);
    put16(0xFFDA); put16(12); put(3); put(1); put(0x00); put(2); put(0x11); put(3); put(0x11); put(0); put(63); put(0);
    res->header_bytes = (uint32_t)len;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K4: entropy coder. One single-wave workgroup per restart interval. Per batch of 64 blocks:
//   1. lane L receives block L (load_batch), 2. every lane Huffman-codes its own block into a private LDS strip
//   (MSB-first 32-bit words), 3. a wave prefix sum of the bit counts gives each lane's position and the strips are
//   OR-merged into a window with funnel shifts, 4. whole words are byte-swapped and streamed to the interval's
//   scratch slot with 256-B coalesced stores. FF bytes are only counted here; stuffing happens in K6.
// ---------------------------------------------------------------------------------------------------------------
constexpr int WIN_WORDS = 1024;  // merge window: 32 Kbit; a typical q95 batch needs ~10 Kbit

struct BitSink {
  uint32_t *buf;   // private strip: word w of lane L at buf[w * 64 + L]
  int lane;
  uint64_t acc;    // low `n` bits valid
  int n, words;
  __device__ __forceinline__ void put(uint32_t bits, int len) {
    acc = (acc << len) | bits;
    n += len;
    if (n >= 32) {
      n -= 32;
      buf[words * 64 + lane] = (uint32_t)(acc >> n);
      words++;
    }
  }
  __device__ __forceinline__ int finish() {  // returns total bit count; final partial word left-aligned, zero filled
    const int total = words * 32 + n;
    if (n > 0) buf[words * 64 + lane] = (uint32_t)(acc << (32 - n));
    return total;
  }
};

__device__ __forceinline__ int ff_bytes(uint32_t x) {
  return __popc(x & 0x80808080u & ((x & 0x7F7F7F7Fu) + 0x01010101u));
}

__global__ __launch_bounds__(64) void k_encode(const Geom g, const int16_t *__restrict__ coef,
                                               const DeviceTables *__restrict__ tab, uint8_t *__restrict__ scratch,
                                               const size_t slot_bytes, uint32_t *__restrict__ seg_bytes,
                                               uint32_t *__restrict__ seg_ff) {
  __shared__ __attribute__((aligned(16))) uint32_t s_buf[MAX_BLOCK_WORDS * 64];  // also the load_batch staging area
  __shared__ uint32_t s_win[WIN_WORDS + 1];
  __shared__ uint32_t s_lut[LUT_SIZE];
  static_assert(sizeof(uint32_t) * MAX_BLOCK_WORDS * 64 >= STAGE_BYTES, "staging must fit in the strip area");
  const int lane = threadIdx.x;
  for (int i = lane; i < LUT_SIZE; i += 64) s_lut[i] = tab->lut[i];
  __syncthreads();

  const long long seg = blockIdx.x;
  const long long seg_first = seg * g.ri * g.bpm;
  const long long seg_end = min(seg_first + (long long)g.ri * g.bpm, g.mcu_count * g.bpm);
  uint32_t *gout = reinterpret_cast<uint32_t *>(scratch + (size_t)seg * slot_bytes);
  long long gw = 0;        // whole words already written to the slot
  uint32_t carry = 0;      // pending partial word (MSB aligned), `cbits` valid bits
  int cbits = 0;
  int ffcount = 0;

  for (long long b0 = seg_first; b0 < seg_end; b0 += 64) {
    const int nvalid = (int)min((long long)64, seg_end - b0);
    uint32_t c[32];
    load_batch(coef + b0 * 64, nvalid, reinterpret_cast<uint8_t *>(s_buf), lane, c);
    int nbits = 0;
    if (lane < nvalid) {
      int chroma;
      const int pred = dc_pred(coef, b0 + lane, seg_first, g.bpm, g.nl, chroma);
      const uint32_t *ldc = s_lut + (chroma ? LUT_DC_C : LUT_DC_L), *lac = s_lut + (chroma ? LUT_AC_C : LUT_AC_L);
      BitSink sk = {s_buf, lane, 0, 0, 0};
      {
        const int diff = coef_at(c, 0) - pred;
        const int nb = nbits_of(diff);
        const uint32_t e = ldc[nb];
        const uint32_t amp = (uint32_t)(diff + (diff >> 31)) & ((1u << nb) - 1u);
        sk.put(((e >> 5) << nb) | amp, (int)(e & 31) + nb);
      }
      int r = 0;
#pragma unroll
      for (int k = 1; k < 64; k++) {
        const int v = coef_at(c, k);
        if (v == 0) { r++; }
        else {
          if (r > 15) {
            const uint32_t z = lac[0xF0];
            do { sk.put(z >> 5, (int)(z & 31)); r -= 16; } while (r > 15);
          }
          const int nb = nbits_of(v);
          const uint32_t e = lac[(r << 4) + nb];
          const uint32_t amp = (uint32_t)(v + (v >> 31)) & ((1u << nb) - 1u);
          sk.put(((e >> 5) << nb) | amp, (int)(e & 31) + nb);
          r = 0;
        }
      }
      if (r > 0) { const uint32_t e = lac[0]; sk.put(e >> 5, (int)(e & 31)); }
      nbits = sk.finish();
    }
    // exclusive prefix sum of bit counts across the wave
    int incl = nbits;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    const int total = __shfl(incl, 63, 64);
    const int my_off = cbits + incl - nbits;      // bit offset relative to the window start of round 0
    const int nw = (nbits + 31) >> 5;             // source words of this lane
    const int sh = my_off & 31, d0 = my_off >> 5;
    const int end_bits = cbits + total;
    __syncthreads();
    for (int rbase = 0; rbase * 32 < end_bits; rbase += WIN_WORDS) {   // rbase in words; one round unless the batch is huge
      const int bits_here = min(end_bits - rbase * 32, WIN_WORDS * 32);
      const int nfull = bits_here >> 5;
      for (int i = lane; i <= nfull; i += 64) s_win[i] = (rbase == 0 && i == 0) ? carry : 0u;
      __syncthreads();
      // dest word d0 + w receives the funnel shift of (src[w-1], src[w]) by sh, for w = 0 .. nw
      uint32_t prev = 0;
      for (int w = 0; w <= nw; w++) {
        const uint32_t cur = w < nw ? s_buf[w * 64 + lane] : 0u;
        const int dd = d0 + w - rbase;
        const uint32_t val = sh ? ((prev << (32 - sh)) | (cur >> sh)) : cur;
        if (dd >= 0 && dd < WIN_WORDS && val) atomicOr(&s_win[dd], val);
        prev = cur;
      }
      __syncthreads();
      for (int i = lane; i < nfull; i += 64) {
        const uint32_t v = s_win[i];
        ffcount += ff_bytes(v);
        gout[gw + i] = __builtin_bswap32(v);
      }
      gw += nfull;
      if (end_bits - rbase * 32 <= WIN_WORDS * 32) {  // last round of this batch: keep the partial word
        carry = (bits_here & 31) ? s_win[nfull] : 0u;
        cbits = bits_here & 31;
      }
      __syncthreads();
    }
  }
  // pad the final byte with 1-bits (T.81 F.1.2.3), emit the tail bytes
  int tail_bytes = (cbits + 7) >> 3;
  if (cbits > 0) {
    const uint32_t padded = carry | (0xFFFFFFFFu >> cbits);
    const uint32_t keep = tail_bytes == 4 ? 0xFFFFFFFFu : ~(0xFFFFFFFFu >> (8 * tail_bytes));
    if (lane == 0) {
      gout[gw] = __builtin_bswap32(padded & keep);
      ffcount += ff_bytes(padded & keep);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) ffcount += __shfl_xor(ffcount, off, 64);
  if (lane == 0) {
    seg_bytes[seg] = (uint32_t)(gw * 4 + tail_bytes);
    seg_ff[seg] = (uint32_t)ffcount;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// K5: exclusive scan of interval sizes (stuffed bytes + 2 marker bytes each). Single workgroup.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_scan(const uint32_t *__restrict__ seg_bytes, const uint32_t *__restrict__ seg_ff,
                                               unsigned long long *__restrict__ seg_off, const long long nseg,
                                               DeviceResult *__restrict__ res) {
  __shared__ unsigned long long s_part[16];
  __shared__ unsigned long long s_carry;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (long long base = 0; base < nseg; base += 1024) {
    const long long i = base + t;
    const unsigned long long v = i < nseg ? (unsigned long long)seg_bytes[i] + seg_ff[i] + 2ull : 0ull;
    unsigned long long incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned long long o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) s_part[wv] = incl;
    __syncthreads();
    unsigned long long wbase = 0;
    for (int k = 0; k < wv; k++) wbase += s_part[k];
    const unsigned long long carry = s_carry;
    if (i < nseg) seg_off[i] = carry + wbase + incl - v;
    __syncthreads();
    if (t == 1023) s_carry = carry + wbase + incl;
    __syncthreads();
  }
  if (t == 0) { res->scan_bytes = s_carry; res->flags = 0; }
}

// ---------------------------------------------------------------------------------------------------------------
// K6: byte stuffing + compaction. One wave per restart interval: reads the interval's un-stuffed bytes from its
// scratch slot 1 KiB at a time, inserts 00 after every FF, writes the result at the interval's final offset and
// appends RSTn (or EOI after the last interval of the image).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_compact(const Geom g, const uint8_t *__restrict__ scratch, const size_t slot_bytes,
                                                const uint32_t *__restrict__ seg_bytes,
                                                const unsigned long long *__restrict__ seg_off, const long long nseg,
                                                uint8_t *__restrict__ out, const size_t capacity,
                                                const DeviceResult *__restrict__ res) {
  __shared__ uint8_t s_stage[2048 + 64];
  const int lane = threadIdx.x;
  const long long seg = blockIdx.x;
  if (res->scan_bytes > capacity) return;  // uniform; the host sees scan_bytes > capacity, grows the buffer, re-runs K6
  const uint8_t *src = scratch + (size_t)seg * slot_bytes;
  const uint32_t nbytes = seg_bytes[seg];
  unsigned long long dpos = seg_off[seg];
  uint8_t *stage = s_stage;
  for (uint32_t base = 0; base < nbytes; base += 1024) {
    const uint32_t o = base + lane * 16;
    const int nv = o < nbytes ? (int)min(16u, nbytes - o) : 0;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (nv > 0) v = *reinterpret_cast<const uint4 *>(src + o);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) cnt += (i < nv && ((w[i >> 2] >> ((i & 3) * 8)) & 255) == 0xFF) ? 1 : 0;
    const int outn = nv + cnt;
    int incl = outn;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    const int total = __shfl(incl, 63, 64);
    int pos = incl - outn;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (i < nv) {
        const uint8_t b = (uint8_t)((w[i >> 2] >> ((i & 3) * 8)) & 255);
        stage[pos++] = b;
        if (b == 0xFF) stage[pos++] = 0;
      }
    }
    __syncthreads();
    for (int i = lane; i < total; i += 64) out[dpos + i] = stage[i];
    dpos += total;
    __syncthreads();
  }
  if (lane == 0) {
    const long long gseg = g.mcu_first / g.ri + seg;
    const bool last = g.last_strip && (seg == nseg - 1);
    out[dpos] = 0xFF;
    out[dpos + 1] = last ? 0xD9 : (uint8_t)(0xD0 + (gseg & 7));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Synthetic image, generated on the device so the bench does not depend on a 1 GB upload.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
__device__ __forceinline__ int tri(int t, int P) { const int u = t % P; return u < P / 2 ? u : P - 1 - u; }

__global__ void k_synth(uint8_t *dst, const int W, const int y0, const int rows, const size_t pitch, const int bgr) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (x >= W || r >= rows) return;
  const int y = y0 + r;
  const int gx = tri(x, 1024), gy = tri(y, 768), gd = tri(x + 2 * y, 320);
  int base[3] = {48 + gx * 96 / 512 + gy * 64 / 384, 40 + gx * 64 / 512 + gd * 96 / 160, 56 + gy * 96 / 384 + gd * 48 / 160};
  const int step = (((x / 208) + (y / 250)) & 1) * 24;
  uint8_t px[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const uint32_t idx = ((uint32_t)y * (uint32_t)W + (uint32_t)x) * 3u + (uint32_t)c;
    const uint32_t h = fmix32((idx * 0x9E3779B1u) ^ 0x4D493335u);
    int n = (int)((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) - 510;
    n >>= 4;
    px[c] = (uint8_t)min(255, max(0, base[c] + step + n));
  }
  uint8_t *p = dst + (size_t)r * pitch + (size_t)x * 3;
  p[0] = bgr ? px[2] : px[0]; p[1] = px[1]; p[2] = bgr ? px[0] : px[2];
}

// ---------------------------------------------------------------------------------------------------------------
// Launchers. hipGetLastError() is per-thread sticky state that other HIP users in the process (e.g. a framework)
// may have left set, so it is cleared before each launch and read right after it.
// ---------------------------------------------------------------------------------------------------------------
#define MIJ_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
template <int HS, int VS>
static hipError_t launch_transform_t(const Geom &g, const TransformArgs &a, int interleaved, int amode, hipStream_t s) {
  constexpr int MPT = TCfg<HS, VS>::MPT;
  const unsigned grid = (unsigned)((g.mcu_count + MPT - 1) / MPT);
  if (interleaved) MIJ_LAUNCH((k_transform<HS, VS, true>), dim3(grid), dim3(256), 0, s, g, a, amode);
  else MIJ_LAUNCH((k_transform<HS, VS, false>), dim3(grid), dim3(256), 0, s, g, a, amode);
  return hipGetLastError();
}

hipError_t launch_transform(const Geom &g, const TransformArgs &a, int interleaved, hipStream_t s) {
  // widest naturally aligned access the caller's pointer / pitch allow
  const uintptr_t al = (uintptr_t)a.src | (uintptr_t)a.pitch | (interleaved ? 0 : (uintptr_t)a.plane_stride);
  const int amode = (al & 7) == 0 ? 8 : ((al & 3) == 0 ? 4 : 1);
  hipError_t e;
  if (g.hs == 1 && g.vs == 1) e = launch_transform_t<1, 1>(g, a, interleaved, amode, s);
  else if (g.hs == 2 && g.vs == 1) e = launch_transform_t<2, 1>(g, a, interleaved, amode, s);
  else if (g.hs == 2 && g.vs == 2) e = launch_transform_t<2, 2>(g, a, interleaved, amode, s);
  else if (g.hs == 1 && g.vs == 2) e = launch_transform_t<1, 2>(g, a, interleaved, amode, s);
  else if (g.hs == 4 && g.vs == 1) e = launch_transform_t<4, 1>(g, a, interleaved, amode, s);
  else if (g.hs == 4 && g.vs == 2) e = launch_transform_t<4, 2>(g, a, interleaved, amode, s);
  else return hipErrorInvalidValue;
  if (e != hipSuccess) return e;
  if (g.mcux * g.hs > g.wib0 || g.mcuy * g.vs > g.hib0) {
    const unsigned grid = (unsigned)((g.mcu_count + 255) / 256);
    MIJ_LAUNCH(k_fix_dummy_dc, dim3(grid), dim3(256), 0, s, g, a.coef);
    e = hipGetLastError();
  }
  return e;
}

static long long num_segments(const Geom &g) { return (g.mcu_count + g.ri - 1) / g.ri; }

hipError_t launch_histogram(const Geom &g, const int16_t *coef, uint32_t *hist, hipStream_t s) {
  MIJ_LAUNCH(k_histogram, dim3((unsigned)num_segments(g)), dim3(64), 0, s, g, coef, hist);
  return hipGetLastError();
}

hipError_t launch_build_tables(const Geom &g, const uint32_t *hist, int optimize, const Quant *qt, DeviceTables *tab,
                               uint8_t *out, DeviceResult *res, hipStream_t s) {
  MIJ_LAUNCH(k_build_tables, dim3(1), dim3(256), 0, s, g, hist, optimize, qt, tab, out, res);
  return hipGetLastError();
}

hipError_t launch_encode(const Geom &g, const int16_t *coef, const DeviceTables *tab, uint8_t *scratch,
                         size_t slot_bytes, uint32_t *seg_bytes, uint32_t *seg_ff, long long nseg, hipStream_t s) {
  MIJ_LAUNCH(k_encode, dim3((unsigned)nseg), dim3(64), 0, s, g, coef, tab, scratch, slot_bytes, seg_bytes, seg_ff);
  return hipGetLastError();
}

hipError_t launch_scan(const uint32_t *seg_bytes, const uint32_t *seg_ff, unsigned long long *seg_off, long long nseg,
                       DeviceResult *res, hipStream_t s) {
  MIJ_LAUNCH(k_scan, dim3(1), dim3(1024), 0, s, seg_bytes, seg_ff, seg_off, nseg, res);
  return hipGetLastError();
}

hipError_t launch_compact(const Geom &g, const uint8_t *scratch, size_t slot_bytes, const uint32_t *seg_bytes,
                          const unsigned long long *seg_off, long long nseg, uint8_t *out_scan, size_t capacity,
                          const DeviceResult *res, hipStream_t s) {
  MIJ_LAUNCH(k_compact, dim3((unsigned)nseg), dim3(64), 0, s, g, scratch, slot_bytes, seg_bytes, seg_off, nseg,
                     out_scan, capacity, res);
  return hipGetLastError();
}

hipError_t launch_synth(uint8_t *dst, int W, int y0, int rows, size_t pitch, int bgr, hipStream_t s) {
  MIJ_LAUNCH(k_synth, dim3((W + 255) / 256, rows), dim3(256), 0, s, dst, W, y0, rows, pitch, bgr);
  return hipGetLastError();
}

}  // namespace mij

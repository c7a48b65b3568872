/*
 * jpeg_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * CPU restatement of the JPEG encode path that the reference delegates to NVIDIA nvJPEG:
 *   reference call site  src/ImageCompressorDll/ImageCompressorImpl.cu:280  nvjpegEncodeImage(...)
 *   parameters           src/ImageCompressorDll/ImageCompressorImpl.cu:28-31 (encoding, optimizedHuffman,
 *                        quality, sampling factors), .cuh:40-45 (defaults 8320x40000, q95, optimise on)
 *   retrieve             src/ImageCompressorDll/ImageCompressorImpl.cu:285-287 nvjpegEncodeRetrieveBitstream
 *
 * PARITY STATUS: **parity unpinned against the reference itself.** The arithmetic lives in nvJPEG
 * (CUDA Toolkit 11.3, closed source, un-vendored: reference CMakeLists.txt:16-20, .gitignore:1-2) and the
 * reference ships no tests, fixtures or golden vectors. What this file IS pinned
 * against, bit for bit, is the published baseline-JPEG algorithm (ITU-T T.81 Annex A/F/K, JFIF 1.02) with the
 * integer choices of libjpeg-turbo 3.1.4.1 (jccolor / jcsample / jfdctint "islow" / jcdctmgr / jchuff) and of
 * IJG libjpeg 9d for the 4:4:0 and 4:1:1 samplings; tests/test_oracle_pin.py compares whole files
 * byte-for-byte with those stock encoders, and tests/golden/ holds the vectors.
 *
 * Nothing here comes from the reference (it contains no JPEG arithmetic). ATTRIBUTION: bit-identity with the stock
 * encoders is only possible by following their integer procedures, so several routines RESTATE code of the Independent
 * JPEG Group's libjpeg and of libjpeg-turbo -- fdct_islow (jfdctint.c: the Loeffler-Ligtenberg-Moschytz factorisation
 * with the same 13-bit FIX_* constants and descale points), the colour conversion and downsampling rounding rules
 * (jccolor.c, jcsample.c), the quantiser (jcdctmgr.c), the optimal-table construction incl. length limiting (jchuff.c
 * jpeg_gen_optimal_table), and the progressive scan script and coder (jcparam.c jpeg_simple_progression, jcphuff.c).
 *   This software is based in part on the work of the Independent JPEG Group.
 *   libjpeg: Copyright (C) 1991-2020, Thomas G. Lane, Guido Vollbeding.
 *   libjpeg-turbo: Copyright (C) 2009-2024 D. R. Commander et al.; distributed under the IJG licence and the
 *   Modified (3-clause) BSD licence (see libjpeg-turbo's LICENSE.md / README.ijg for the full terms: the IJG
 *   licence permits use, copying, modification and distribution provided this acknowledgement is retained).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#define MJO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------
 * Synthetic image generator, SURVEY.md section 8(d) (normative spec; integer only, strip addressable).
 * Produces RGB8 interleaved rows [y0, y0+rows) of a W-wide image.
 * ---------------------------------------------------------------------------------------------- */
static inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h;
}
static inline int tri(int t, int P) { int u = t % P; return u < P / 2 ? u : P - 1 - u; }
static inline int clamp255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

MJO_API void mjo_synth_rgb(uint8_t *dst, int W, int y0, int rows, size_t stride) {
  for (int r = 0; r < rows; r++) {
    int y = y0 + r;
    uint8_t *p = dst + (size_t)r * stride;
    int gy = tri(y, 768);
    for (int x = 0; x < W; x++) {
      int gx = tri(x, 1024), gd = tri(x + 2 * y, 320);
      int base[3];
      base[0] = 48 + gx * 96 / 512 + gy * 64 / 384;
      base[1] = 40 + gx * 64 / 512 + gd * 96 / 160;
      base[2] = 56 + gy * 96 / 384 + gd * 48 / 160;
      int step = (((x / 208) + (y / 250)) & 1) * 24;
      for (int c = 0; c < 3; c++) {
        uint32_t idx = ((uint32_t)y * (uint32_t)W + (uint32_t)x) * 3u + (uint32_t)c;
        uint32_t h = fmix32((idx * 0x9E3779B1u) ^ 0x4D493335u);
        int n = (int)((h & 255) + ((h >> 8) & 255) + ((h >> 16) & 255) + (h >> 24)) - 510;
        n >>= 4; /* arithmetic shift */
        p[x * 3 + c] = (uint8_t)clamp255(base[c] + step + n);
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Tables: T.81 Annex K.1 quantisation, K.3 Huffman; zig-zag order (Figure A.6).
 * ---------------------------------------------------------------------------------------------- */
static const uint8_t k_std_lum_q[64] = {
  16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
  14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
  49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t k_std_chr_q[64] = {
  17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
  47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
/* zz -> natural index */
static const uint8_t k_zigzag[64] = {
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
  28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
  47, 55, 62, 63};

static const uint8_t k_dc_lum_bits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t k_dc_chr_bits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t k_dc_vals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t k_ac_lum_bits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t k_ac_lum_vals[162] = {
  0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
  0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
  0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
  0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
  0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
  0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
  0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
  0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
  0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t k_ac_chr_bits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t k_ac_chr_vals[162] = {
  0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
  0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
  0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
  0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
  0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
  0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
  0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
  0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
  0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

/* IJG quality scaling (the convention nvjpegEncoderParamsSetQuality documents; reference .cu:30).
 * which: 0 = luminance, 1 = chrominance. Output in natural (row-major) order, clamped to 1..255. */
MJO_API void mjo_quant_table(int quality, int which, uint16_t out[64]) {
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  const uint8_t *base = which ? k_std_chr_q : k_std_lum_q;
  for (int i = 0; i < 64; i++) {
    long t = ((long)base[i] * scale + 50L) / 100L;
    if (t <= 0) t = 1;
    if (t > 255) t = 255; /* force_baseline */
    out[i] = (uint16_t)t;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Geometry. css uses the nvjpegChromaSubsampling_t integer values:
 *   0=444 1=422 2=420 3=440 4=411 5=410.  Chroma is always 1x1; luma is hs x vs.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int W, H, hs, vs;
  int mcux, mcuy;       /* MCUs across / down */
  int bpm;              /* blocks per MCU = hs*vs + 2 */
  int cw[3], ch[3];     /* component sample dims (downsampled, unpadded) */
  int wib[3], hib[3];   /* width / height in blocks (real blocks) */
} geom_t;

static int css_factors(int css, int *hs, int *vs) {
  switch (css) {
    case 0: *hs = 1; *vs = 1; return 0;
    case 1: *hs = 2; *vs = 1; return 0;
    case 2: *hs = 2; *vs = 2; return 0;
    case 3: *hs = 1; *vs = 2; return 0;
    case 4: *hs = 4; *vs = 1; return 0;
    case 5: *hs = 4; *vs = 2; return 0;
    default: return -1;
  }
}
static int ceil_div(int a, int b) { return (a + b - 1) / b; }

static int make_geom(geom_t *g, int W, int H, int css) {
  if (W <= 0 || H <= 0 || W > 65535 || H > 65535) return -1;
  if (css_factors(css, &g->hs, &g->vs)) return -1;
  g->W = W; g->H = H;
  g->mcux = ceil_div(W, 8 * g->hs);
  g->mcuy = ceil_div(H, 8 * g->vs);
  g->bpm = g->hs * g->vs + 2;
  for (int c = 0; c < 3; c++) {
    int h = c == 0 ? g->hs : 1, v = c == 0 ? g->vs : 1;
    g->cw[c] = ceil_div(W * h, g->hs);
    g->ch[c] = ceil_div(H * v, g->vs);
    g->wib[c] = ceil_div(g->cw[c], 8);
    g->hib[c] = ceil_div(g->ch[c], 8);
  }
  return 0;
}

MJO_API int mjo_geometry(int W, int H, int css, int out[8]) {
  geom_t g;
  if (make_geom(&g, W, H, css)) return -1;
  out[0] = g.hs; out[1] = g.vs; out[2] = g.mcux; out[3] = g.mcuy; out[4] = g.bpm;
  out[5] = g.wib[0]; out[6] = g.hib[0]; out[7] = g.wib[1];
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Colour conversion: JFIF BT.601 full range, 16-bit fixed point, libjpeg's rounding
 * (Y rounds to nearest; Cb/Cr use 0.5-epsilon so that 255.5 cannot occur).
 * ---------------------------------------------------------------------------------------------- */
#define FIXC(x) ((int32_t)((x) * 65536.0 + 0.5))
static inline void rgb_to_ycc(int r, int g, int b, uint8_t *y, uint8_t *cb, uint8_t *cr) {
  const int32_t half = 1 << 15, off = 128 << 16;
  *y = (uint8_t)((FIXC(0.29900) * r + FIXC(0.58700) * g + FIXC(0.11400) * b + half) >> 16);
  *cb = (uint8_t)((-FIXC(0.16874) * r - FIXC(0.33126) * g + FIXC(0.50000) * b + off + half - 1) >> 16);
  *cr = (uint8_t)((FIXC(0.50000) * r - FIXC(0.41869) * g - FIXC(0.08131) * b + off + half - 1) >> 16);
}

MJO_API void mjo_rgb_to_ycc(const uint8_t *rgb, size_t npix, uint8_t *ycc) {
  for (size_t i = 0; i < npix; i++) rgb_to_ycc(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], &ycc[3 * i], &ycc[3 * i + 1], &ycc[3 * i + 2]);
}

/* pixfmt: 0 = RGB interleaved, 1 = BGR interleaved (cv::Mat CV_8UC3, reference main.cpp:37),
 *         2 = planar R,G,B (3 planes of `stride`*H), 3 = planar B,G,R (nvjpegImage_t + NVJPEG_INPUT_BGR, .cuh:62) */
static void fetch_rgb(const uint8_t *src, size_t stride, int H, int pixfmt, int x, int y, int *r, int *g, int *b) {
  if (pixfmt == 0 || pixfmt == 1) {
    const uint8_t *p = src + (size_t)y * stride + (size_t)x * 3;
    if (pixfmt == 0) { *r = p[0]; *g = p[1]; *b = p[2]; } else { *b = p[0]; *g = p[1]; *r = p[2]; }
  } else {
    size_t plane = stride * (size_t)H, o = (size_t)y * stride + (size_t)x;
    int a = src[o], m = src[plane + o], z = src[2 * plane + o];
    if (pixfmt == 2) { *r = a; *g = m; *b = z; } else { *b = a; *g = m; *r = z; }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Forward DCT: the "accurate integer" (Loeffler-Ligtenberg-Moschytz) 8x8 DCT with 13-bit constants,
 * 2 extra bits kept after the row pass; output is scaled up by 8 (removed by the quantiser).
 * ---------------------------------------------------------------------------------------------- */
#define CONST_BITS 13
#define PASS1_BITS 2
#define F_0_298631336 2446
#define F_0_390180644 3196
#define F_0_541196100 4433
#define F_0_765366865 6270
#define F_0_899976223 7373
#define F_1_175875602 9633
#define F_1_501321110 12299
#define F_1_847759065 15137
#define F_1_961570560 16069
#define F_2_053119869 16819
#define F_2_562915447 20995
#define F_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))

static void fdct_islow(int32_t *d) {
  int32_t t0, t1, t2, t3, t4, t5, t6, t7, t10, t11, t12, t13, z1, z2, z3, z4, z5;
  for (int r = 0; r < 8; r++) {
    int32_t *p = d + r * 8;
    t0 = p[0] + p[7]; t7 = p[0] - p[7]; t1 = p[1] + p[6]; t6 = p[1] - p[6];
    t2 = p[2] + p[5]; t5 = p[2] - p[5]; t3 = p[3] + p[4]; t4 = p[3] - p[4];
    t10 = t0 + t3; t13 = t0 - t3; t11 = t1 + t2; t12 = t1 - t2;
    p[0] = (t10 + t11) << PASS1_BITS;
    p[4] = (t10 - t11) << PASS1_BITS;
    z1 = (t12 + t13) * F_0_541196100;
    p[2] = DESCALE(z1 + t13 * F_0_765366865, CONST_BITS - PASS1_BITS);
    p[6] = DESCALE(z1 + t12 * (-F_1_847759065), CONST_BITS - PASS1_BITS);
    z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
    z5 = (z3 + z4) * F_1_175875602;
    t4 *= F_0_298631336; t5 *= F_2_053119869; t6 *= F_3_072711026; t7 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    p[7] = DESCALE(t4 + z1 + z3, CONST_BITS - PASS1_BITS);
    p[5] = DESCALE(t5 + z2 + z4, CONST_BITS - PASS1_BITS);
    p[3] = DESCALE(t6 + z2 + z3, CONST_BITS - PASS1_BITS);
    p[1] = DESCALE(t7 + z1 + z4, CONST_BITS - PASS1_BITS);
  }
  for (int c = 0; c < 8; c++) {
    int32_t *p = d + c;
    t0 = p[0] + p[56]; t7 = p[0] - p[56]; t1 = p[8] + p[48]; t6 = p[8] - p[48];
    t2 = p[16] + p[40]; t5 = p[16] - p[40]; t3 = p[24] + p[32]; t4 = p[24] - p[32];
    t10 = t0 + t3; t13 = t0 - t3; t11 = t1 + t2; t12 = t1 - t2;
    p[0] = DESCALE(t10 + t11, PASS1_BITS);
    p[32] = DESCALE(t10 - t11, PASS1_BITS);
    z1 = (t12 + t13) * F_0_541196100;
    p[16] = DESCALE(z1 + t13 * F_0_765366865, CONST_BITS + PASS1_BITS);
    p[48] = DESCALE(z1 + t12 * (-F_1_847759065), CONST_BITS + PASS1_BITS);
    z1 = t4 + t7; z2 = t5 + t6; z3 = t4 + t6; z4 = t5 + t7;
    z5 = (z3 + z4) * F_1_175875602;
    t4 *= F_0_298631336; t5 *= F_2_053119869; t6 *= F_3_072711026; t7 *= F_1_501321110;
    z1 *= -F_0_899976223; z2 *= -F_2_562915447; z3 *= -F_1_961570560; z4 *= -F_0_390180644;
    z3 += z5; z4 += z5;
    p[56] = DESCALE(t4 + z1 + z3, CONST_BITS + PASS1_BITS);
    p[40] = DESCALE(t5 + z2 + z4, CONST_BITS + PASS1_BITS);
    p[24] = DESCALE(t6 + z2 + z3, CONST_BITS + PASS1_BITS);
    p[8] = DESCALE(t7 + z1 + z4, CONST_BITS + PASS1_BITS);
  }
}

/* Quantise one DCT output (scaled by 8): round half away from zero of coef / (8*q). */
static inline int quantise(int32_t v, int q) {
  int32_t d = q << 3;
  if (v < 0) return -(int)(((-v) + (d >> 1)) / d);
  return (int)((v + (d >> 1)) / d);
}

/* ------------------------------------------------------------------------------------------------
 * Stage A on the CPU: whole image -> quantised coefficients in MCU order, each block in zig-zag order.
 * coef must hold mcux*mcuy*bpm*64 int16. This is the layout the HIP path uses in HBM.
 * ---------------------------------------------------------------------------------------------- */
static uint8_t *build_component_plane(const uint8_t *full, int W, int H, const geom_t *g, int c, int *pw_out,
                                      int *ph_out) {
  /* full: W x H plane (uint8). Returns the downsampled, edge-padded component plane:
   * width = wib*8, height = mcuy * v_c * 8 (rows past the real data are replicas; they are only read by
   * blocks that libjpeg also treats as real). Padding follows jcsample/jcprepct: the input is widened by
   * replicating its last column, its last partial row group is completed by replicating the last input row,
   * and after downsampling the last output row is replicated downwards. */
  int hexp = c == 0 ? 1 : g->hs, vexp = c == 0 ? 1 : g->vs;
  int vcomp = c == 0 ? g->vs : 1;
  int pw = g->wib[c] * 8, ph = g->mcuy * vcomp * 8;
  uint8_t *out = (uint8_t *)malloc((size_t)pw * ph);
  if (!out) return NULL;
  int in_w = pw * hexp;                       /* padded input width */
  int in_h = ceil_div(H, g->vs) * g->vs;      /* input rows after completing the last row group */
  int out_rows = in_h / vexp;                 /* output rows produced by real downsampling */
  uint8_t *rows = (uint8_t *)malloc((size_t)in_w * vexp);
  for (int oy = 0; oy < out_rows; oy++) {
    for (int v = 0; v < vexp; v++) {
      int iy = oy * vexp + v;
      if (iy >= H) iy = H - 1;
      const uint8_t *s = full + (size_t)iy * W;
      uint8_t *d = rows + (size_t)v * in_w;
      int n = W < in_w ? W : in_w;
      memcpy(d, s, (size_t)n);
      for (int x = n; x < in_w; x++) d[x] = s[W - 1];
    }
    uint8_t *o = out + (size_t)oy * pw;
    if (hexp == 1 && vexp == 1) {
      memcpy(o, rows, (size_t)pw);
    } else if (hexp == 2 && vexp == 1) { /* alternating bias 0,1,0,1 */
      int bias = 0;
      for (int x = 0; x < pw; x++) { o[x] = (uint8_t)((rows[2 * x] + rows[2 * x + 1] + bias) >> 1); bias ^= 1; }
    } else if (hexp == 2 && vexp == 2) { /* alternating bias 1,2,1,2 */
      int bias = 1;
      const uint8_t *r0 = rows, *r1 = rows + in_w;
      for (int x = 0; x < pw; x++) {
        o[x] = (uint8_t)((r0[2 * x] + r0[2 * x + 1] + r1[2 * x] + r1[2 * x + 1] + bias) >> 2);
        bias ^= 3;
      }
    } else { /* general box filter, round half up */
      int numpix = hexp * vexp, half = numpix / 2;
      for (int x = 0; x < pw; x++) {
        int s = 0;
        for (int v = 0; v < vexp; v++)
          for (int h = 0; h < hexp; h++) s += rows[(size_t)v * in_w + x * hexp + h];
        o[x] = (uint8_t)((s + half) / numpix);
      }
    }
  }
  for (int oy = out_rows; oy < ph; oy++) memcpy(out + (size_t)oy * pw, out + (size_t)(out_rows - 1) * pw, (size_t)pw);
  free(rows);
  *pw_out = pw; *ph_out = ph;
  return out;
}

MJO_API int mjo_coefficients(const uint8_t *src, int W, int H, size_t stride, int pixfmt, int quality, int css,
                             int16_t *coef) {
  geom_t g;
  if (make_geom(&g, W, H, css)) return -1;
  uint16_t qt[2][64];
  mjo_quant_table(quality, 0, qt[0]);
  mjo_quant_table(quality, 1, qt[1]);
  size_t np = (size_t)W * H;
  uint8_t *planes = (uint8_t *)malloc(np * 3);
  if (!planes) return -2;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int r, gg, b;
      fetch_rgb(src, stride, H, pixfmt, x, y, &r, &gg, &b);
      size_t o = (size_t)y * W + x;
      rgb_to_ycc(r, gg, b, &planes[o], &planes[np + o], &planes[2 * np + o]);
    }
  for (int c = 0; c < 3; c++) {
    int pw, ph;
    uint8_t *cp = build_component_plane(planes + np * c, W, H, &g, c, &pw, &ph);
    if (!cp) { free(planes); return -2; }
    int hc = c == 0 ? g.hs : 1, vc = c == 0 ? g.vs : 1;
    const uint16_t *q = qt[c ? 1 : 0];
    int blk_off = c == 0 ? 0 : g.hs * g.vs + (c - 1);
    for (int my = 0; my < g.mcuy; my++)
      for (int mx = 0; mx < g.mcux; mx++)
        for (int yi = 0; yi < vc; yi++)
          for (int xi = 0; xi < hc; xi++) {
            int by = my * vc + yi, bx = mx * hc + xi;
            int16_t *o = coef + (((size_t)my * g.mcux + mx) * g.bpm + blk_off + (c == 0 ? yi * hc + xi : 0)) * 64;
            if (by < g.hib[c] && bx < g.wib[c]) {
              int32_t d[64];
              for (int y = 0; y < 8; y++)
                for (int x = 0; x < 8; x++) d[y * 8 + x] = (int32_t)cp[(size_t)(by * 8 + y) * pw + bx * 8 + x] - 128;
              fdct_islow(d);
              for (int k = 0; k < 64; k++) { int n = k_zigzag[k]; o[k] = (int16_t)quantise(d[n], q[n]); }
            } else {
              /* dummy block (T.81 A.2.4 padding to whole MCUs): AC = 0, DC copied so that its DC difference
               * is zero -- from the block to its left, or for a dummy bottom row from the last block of the
               * row above within the same MCU. */
              memset(o, 0, 128);
              if (by < g.hib[c]) o[0] = o[-64];                 /* right edge: previous block of this row */
              else o[0] = (o - (size_t)(xi + 1) * 64)[0];       /* bottom: last block of the row above in this MCU */
            }
          }
    free(cp);
  }
  free(planes);
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Huffman tables.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t bits[17]; uint8_t vals[256]; int nvals; uint16_t code[256]; uint8_t len[256]; } htab_t;

static void derive_codes(htab_t *t) { /* T.81 Annex C */
  memset(t->code, 0, sizeof t->code);
  memset(t->len, 0, sizeof t->len);
  int p = 0; unsigned code = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < t->bits[l]; i++) { t->code[t->vals[p]] = (uint16_t)code; t->len[t->vals[p]] = (uint8_t)l; p++; code++; }
    code <<= 1;
  }
  t->nvals = p;
}
static void std_table(htab_t *t, int is_ac, int is_chroma) {
  memset(t, 0, sizeof *t);
  if (!is_ac) {
    memcpy(t->bits, is_chroma ? k_dc_chr_bits : k_dc_lum_bits, 17);
    memcpy(t->vals, k_dc_vals, 12);
  } else {
    memcpy(t->bits, is_chroma ? k_ac_chr_bits : k_ac_lum_bits, 17);
    memcpy(t->vals, is_chroma ? k_ac_chr_vals : k_ac_lum_vals, 162);
  }
  derive_codes(t);
}

/* Optimal table from symbol frequencies: T.81 K.2 (Huffman's procedure with the reserved all-ones code point
 * via pseudo-symbol 256, ties resolved towards the larger symbol value) + the Figure K.3 adjustment that
 * limits code lengths to 16. freq has 257 entries; freq[256] is overwritten. */
MJO_API int mjo_gen_optimal_table(const uint32_t freq_in[257], uint8_t bits_out[17], uint8_t vals_out[256]) {
  enum { MAXLEN = 32 };
  long freq[257]; int codesize[257], others[257]; uint8_t bits[MAXLEN + 1];
  memset(bits, 0, sizeof bits);
  for (int i = 0; i < 257; i++) { freq[i] = (long)freq_in[i]; codesize[i] = 0; others[i] = -1; }
  freq[256] = 1;
  for (;;) {
    int c1 = -1, c2 = -1; long v = 1000000000L;
    for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
    v = 1000000000L;
    for (int i = 0; i <= 256; i++) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
    if (c2 < 0) break;
    freq[c1] += freq[c2]; freq[c2] = 0;
    codesize[c1]++; while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
    others[c1] = c2;
    codesize[c2]++; while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
  }
  for (int i = 0; i <= 256; i++) if (codesize[i]) { if (codesize[i] > MAXLEN) return -1; bits[codesize[i]]++; }
  int i;
  for (i = MAXLEN; i > 16; i--)
    while (bits[i] > 0) {
      int j = i - 2; while (bits[j] == 0) j--;
      bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
    }
  while (bits[i] == 0) i--;
  bits[i]--;
  memcpy(bits_out, bits, 17);
  bits_out[0] = 0;
  int p = 0;
  memset(vals_out, 0, 256);
  for (int l = 1; l <= MAXLEN; l++)
    for (int j = 0; j <= 255; j++) if (codesize[j] == l) vals_out[p++] = (uint8_t)j;
  return p;
}

static inline int nbits_of(int v) { int n = 0; if (v < 0) v = -v; while (v) { n++; v >>= 1; } return n; }

/* Symbol statistics of the coefficient stream exactly as the entropy coder will see it (DC prediction resets at
 * every restart interval). hist = 4 x 257 uint32: [0]=DC luma [1]=AC luma [2]=DC chroma [3]=AC chroma. */
MJO_API int mjo_histogram(const int16_t *coef, int W, int H, int css, int restart_interval, uint32_t *hist) {
  geom_t g;
  if (make_geom(&g, W, H, css)) return -1;
  memset(hist, 0, 4 * 257 * sizeof(uint32_t));
  long nmcu = (long)g.mcux * g.mcuy;
  int pred[3] = {0, 0, 0};
  int nl = g.hs * g.vs;
  for (long m = 0; m < nmcu; m++) {
    if (restart_interval && m % restart_interval == 0) pred[0] = pred[1] = pred[2] = 0;
    for (int b = 0; b < g.bpm; b++) {
      int c = b < nl ? 0 : b - nl + 1;
      const int16_t *blk = coef + ((size_t)m * g.bpm + b) * 64;
      uint32_t *hdc = hist + (c ? 2 : 0) * 257, *hac = hist + (c ? 3 : 1) * 257;
      hdc[nbits_of(blk[0] - pred[c])]++;
      pred[c] = blk[0];
      int r = 0;
      for (int k = 1; k < 64; k++) {
        if (blk[k] == 0) { r++; continue; }
        while (r > 15) { hac[0xF0]++; r -= 16; }
        hac[(r << 4) + nbits_of(blk[k])]++;
        r = 0;
      }
      if (r > 0) hac[0]++;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Bit writer with FF00 stuffing, marker writer, entropy coder (T.81 Annex F.1.2).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t *p; size_t n, cap; uint32_t acc; int nacc; } bw_t;
static int bw_reserve(bw_t *w, size_t extra) {
  if (w->n + extra <= w->cap) return 0;
  size_t nc = w->cap ? w->cap * 2 : 1 << 16;
  while (nc < w->n + extra) nc *= 2;
  uint8_t *q = (uint8_t *)realloc(w->p, nc);
  if (!q) return -1;
  w->p = q; w->cap = nc; return 0;
}
static void bw_byte(bw_t *w, int b) { if (bw_reserve(w, 1)) return; w->p[w->n++] = (uint8_t)b; }
static void bw_u16(bw_t *w, int v) { bw_byte(w, v >> 8); bw_byte(w, v & 255); }
static void bw_bits(bw_t *w, uint32_t code, int len) {
  if (!len) return;
  w->acc = (w->acc << len) | (code & ((1u << len) - 1));
  w->nacc += len;
  while (w->nacc >= 8) {
    int b = (w->acc >> (w->nacc - 8)) & 255;
    bw_byte(w, b);
    if (b == 0xFF) bw_byte(w, 0);
    w->nacc -= 8;
  }
}
static void bw_flush(bw_t *w) { if (w->nacc) bw_bits(w, 0x7F, 8 - w->nacc); w->acc = 0; w->nacc = 0; }

static void encode_block(bw_t *w, const int16_t *blk, int *pred, const htab_t *dc, const htab_t *ac) {
  int diff = blk[0] - *pred; *pred = blk[0];
  int t = diff, t2 = diff;
  if (t < 0) { t = -t; t2--; }
  int n = nbits_of(t);
  bw_bits(w, dc->code[n], dc->len[n]);
  if (n) bw_bits(w, (uint32_t)t2, n);
  int r = 0;
  for (int k = 1; k < 64; k++) {
    int v = blk[k];
    if (v == 0) { r++; continue; }
    while (r > 15) { bw_bits(w, ac->code[0xF0], ac->len[0xF0]); r -= 16; }
    t = v; t2 = v;
    if (t < 0) { t = -t; t2--; }
    n = nbits_of(t);
    int s = (r << 4) + n;
    bw_bits(w, ac->code[s], ac->len[s]);
    bw_bits(w, (uint32_t)t2, n);
    r = 0;
  }
  if (r > 0) bw_bits(w, ac->code[0], ac->len[0]);
}

static void write_dht(bw_t *w, const htab_t *t, int tc_th) {
  bw_u16(w, 0xFFC4); bw_u16(w, 2 + 1 + 16 + t->nvals); bw_byte(w, tc_th);
  for (int i = 1; i <= 16; i++) bw_byte(w, t->bits[i]);
  for (int i = 0; i < t->nvals; i++) bw_byte(w, t->vals[i]);
}

/* Entropy-code a coefficient buffer (MCU order / zig-zag, as produced by mjo_coefficients or by the HIP path).
 * tables_io (optional, 4 x (17 + 256) bytes: DC luma, AC luma, DC chroma, AC chroma, each bits[17] then vals[256]):
 * when optimize != 0 the generated tables are written there; when optimize == 0 and tables_io != NULL and
 * tables_io[0] == 0xFF... not used -- fixed tables are always Annex K.3.
 * headers: 1 = complete JFIF file (marker order SOI APP0 DQT DQT SOF0 DHTx4 [DRI] SOS ... EOI), 0 = entropy-coded
 * segment only (no markers except RSTn). */
static int encode_worker(const int16_t *coef, int W, int H, int frame_h, int quality, int css, int optimize,
                         int restart_interval, int headers, const uint32_t *hist_override, int rst_start,
                         int trailing_rst, uint8_t *tables_out, uint8_t **out, size_t *out_len) {
  geom_t g;
  if (make_geom(&g, W, H, css)) return -1;
  if (restart_interval < 0 || restart_interval > 65535) return -1;
  htab_t tab[4];
  if (optimize) {
    uint32_t *hist = (uint32_t *)malloc(4 * 257 * sizeof(uint32_t));
    if (hist_override) memcpy(hist, hist_override, 4 * 257 * sizeof(uint32_t));
    else mjo_histogram(coef, W, H, css, restart_interval, hist);
    for (int i = 0; i < 4; i++) {
      memset(&tab[i], 0, sizeof tab[i]);
      if (mjo_gen_optimal_table(hist + i * 257, tab[i].bits, tab[i].vals) < 0) { free(hist); return -3; }
      derive_codes(&tab[i]);
    }
    free(hist);
  } else {
    std_table(&tab[0], 0, 0); std_table(&tab[1], 1, 0); std_table(&tab[2], 0, 1); std_table(&tab[3], 1, 1);
  }
  if (tables_out)
    for (int i = 0; i < 4; i++) { memcpy(tables_out + i * 273, tab[i].bits, 17); memcpy(tables_out + i * 273 + 17, tab[i].vals, 256); }

  bw_t w; memset(&w, 0, sizeof w);
  if (headers & 1) {
    uint16_t qt[2][64];
    mjo_quant_table(quality, 0, qt[0]); mjo_quant_table(quality, 1, qt[1]);
    bw_u16(&w, 0xFFD8);
    bw_u16(&w, 0xFFE0); bw_u16(&w, 16);
    bw_byte(&w, 'J'); bw_byte(&w, 'F'); bw_byte(&w, 'I'); bw_byte(&w, 'F'); bw_byte(&w, 0);
    bw_byte(&w, 1); bw_byte(&w, 1); bw_byte(&w, 0); bw_u16(&w, 1); bw_u16(&w, 1); bw_byte(&w, 0); bw_byte(&w, 0);
    for (int t = 0; t < 2; t++) {
      bw_u16(&w, 0xFFDB); bw_u16(&w, 67); bw_byte(&w, t);
      for (int k = 0; k < 64; k++) bw_byte(&w, qt[t][k_zigzag[k]]);
    }
    bw_u16(&w, 0xFFC0); bw_u16(&w, 17); bw_byte(&w, 8); bw_u16(&w, frame_h); bw_u16(&w, W); bw_byte(&w, 3);
    bw_byte(&w, 1); bw_byte(&w, (g.hs << 4) | g.vs); bw_byte(&w, 0);
    bw_byte(&w, 2); bw_byte(&w, 0x11); bw_byte(&w, 1);
    bw_byte(&w, 3); bw_byte(&w, 0x11); bw_byte(&w, 1);
    write_dht(&w, &tab[0], 0x00); write_dht(&w, &tab[1], 0x10); write_dht(&w, &tab[2], 0x01); write_dht(&w, &tab[3], 0x11);
    if (restart_interval) { bw_u16(&w, 0xFFDD); bw_u16(&w, 4); bw_u16(&w, restart_interval); }
    bw_u16(&w, 0xFFDA); bw_u16(&w, 12); bw_byte(&w, 3);
    bw_byte(&w, 1); bw_byte(&w, 0x00); bw_byte(&w, 2); bw_byte(&w, 0x11); bw_byte(&w, 3); bw_byte(&w, 0x11);
    bw_byte(&w, 0); bw_byte(&w, 63); bw_byte(&w, 0);
  }
  long nmcu = (long)g.mcux * g.mcuy;
  int pred[3] = {0, 0, 0}, nl = g.hs * g.vs, rst = rst_start & 7;
  for (long m = 0; m < nmcu; m++) {
    if (restart_interval && m && m % restart_interval == 0) {
      bw_flush(&w); bw_byte(&w, 0xFF); bw_byte(&w, 0xD0 + rst); rst = (rst + 1) & 7;
      pred[0] = pred[1] = pred[2] = 0;
    }
    for (int b = 0; b < g.bpm; b++) {
      int c = b < nl ? 0 : b - nl + 1;
      encode_block(&w, coef + ((size_t)m * g.bpm + b) * 64, &pred[c], &tab[c ? 2 : 0], &tab[c ? 3 : 1]);
    }
  }
  bw_flush(&w);
  if (trailing_rst) { bw_byte(&w, 0xFF); bw_byte(&w, 0xD0 + rst); }
  if (headers & 2) bw_u16(&w, 0xFFD9);
  *out = w.p; *out_len = w.n;
  return 0;
}

MJO_API int mjo_encode_coefficients(const int16_t *coef, int W, int H, int quality, int css, int optimize,
                                    int restart_interval, int headers, uint8_t *tables_out, uint8_t **out,
                                    size_t *out_len) {
  return encode_worker(coef, W, H, H, quality, css, optimize, restart_interval, headers ? 3 : 0, NULL, 0, 0, tables_out, out,
                       out_len);
}

/* One strip of MCU rows of a taller frame (multi-GPU sharding): `coef` holds the strip's MCUs, the strip
 * starts on a restart-interval boundary whose global index is rst_first, and the Huffman tables come from the
 * statistics of the WHOLE frame (hist, 4 x 257, as all-reduced across ranks). parts: bit 0 = emit the frame header
 * (first strip), bit 1 = emit EOI (last strip); a strip that is not the last ends with the RSTn that separates it
 * from the next one. Concatenating the strips' outputs in order gives the file mjo_encode() writes. */
MJO_API int mjo_encode_strip(const int16_t *coef, int W, int strip_h, int frame_h, int quality, int css, int optimize,
                             int restart_interval, const uint32_t *hist, long rst_first, int parts, uint8_t **out,
                             size_t *out_len) {
  return encode_worker(coef, W, strip_h, frame_h, quality, css, optimize, restart_interval, parts & 3, optimize ? hist : NULL,
                       (int)(rst_first & 7), (parts & 2) ? 0 : 1, NULL, out, out_len);
}

/* Whole path: what nvjpegEncodeImage + nvjpegEncodeRetrieveBitstream produce (reference .cu:280-287), as a
 * baseline sequential (SOF0) file. */
MJO_API int mjo_encode(const uint8_t *src, int W, int H, size_t stride, int pixfmt, int quality, int css,
                       int optimize, int restart_interval, uint8_t **out, size_t *out_len) {
  geom_t g;
  if (make_geom(&g, W, H, css)) return -1;
  size_t n = (size_t)g.mcux * g.mcuy * g.bpm * 64;
  int16_t *coef = (int16_t *)malloc(n * sizeof(int16_t));
  if (!coef) return -2;
  int rc = mjo_coefficients(src, W, H, stride, pixfmt, quality, css, coef);
  if (!rc) rc = mjo_encode_coefficients(coef, W, H, quality, css, optimize, restart_interval, 1, NULL, out, out_len);
  free(coef);
  return rc;
}

MJO_API void mjo_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------------
 * Progressive mode (SOF2) -- the mode the reference's encoder is configured for
 * (nvjpegEncoderParamsSetEncoding(NVJPEG_ENCODING_PROGRESSIVE_DCT_HUFFMAN), reference ImageCompressorImpl.cu:28).
 * nvJPEG's scan script is not public; this restates what libjpeg writes for a 3-component YCbCr image:
 * the scan script of jpeg_simple_progression (jcparam.c) and the entropy coding procedures of T.81 G.1.2 as
 * implemented by jcphuff.c (EOB runs, correction bits buffered behind an EOB run, the 0x7FFF / 937-bit flush
 * rules), with an optimal Huffman table per scan (progressive libjpeg always optimises) and the marker order of
 * jcmarker.c (DHT for the tables a scan uses, DRI before the first SOS, SOS). Pinned byte-for-byte against
 * libjpeg-turbo 3.1.4.1 (Pillow, progressive=True) in tests/test_oracle_pin.py.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int ncomp, comp[3], Ss, Se, Ah, Al; } pscan_t;
static const pscan_t k_simple_progression[10] = {
    {3, {0, 1, 2}, 0, 0, 0, 1},                                       /* initial DC scan */
    {1, {0, 0, 0}, 1, 5, 0, 2},                                       /* some luma AC in a hurry */
    {1, {2, 0, 0}, 1, 63, 0, 1}, {1, {1, 0, 0}, 1, 63, 0, 1},         /* chroma AC: Cr, then Cb */
    {1, {0, 0, 0}, 6, 63, 0, 2},                                      /* rest of the luma spectrum */
    {1, {0, 0, 0}, 1, 63, 2, 1},                                      /* next bit of luma AC */
    {3, {0, 1, 2}, 0, 0, 1, 0},                                       /* last DC bit */
    {1, {2, 0, 0}, 1, 63, 1, 0}, {1, {1, 0, 0}, 1, 63, 1, 0},         /* last chroma AC bit */
    {1, {0, 0, 0}, 1, 63, 1, 0}};                                     /* last luma AC bit */

#define MJO_MAX_CORR_BITS 1000
typedef struct {
  bw_t *w;                 /* NULL while gathering statistics */
  uint32_t *count[2];      /* statistics: [table id 0/1][257] */
  const htab_t *tab[2];    /* tables of this scan, by table id */
  unsigned EOBRUN, BE;
  uint8_t corr[MJO_MAX_CORR_BITS];
  int ac_tbl;              /* table id of the (single) component of an AC scan */
} penc_t;

static void pe_symbol(penc_t *e, int tbl, int sym) {
  if (!e->w) e->count[tbl][sym]++;
  else bw_bits(e->w, e->tab[tbl]->code[sym], e->tab[tbl]->len[sym]);
}
static void pe_bits(penc_t *e, unsigned v, int n) { if (e->w && n) bw_bits(e->w, v, n); }
static void pe_eobrun(penc_t *e) {
  if (e->EOBRUN > 0) {
    int nb = 0; unsigned t = e->EOBRUN; while (t > 1) { nb++; t >>= 1; }
    pe_symbol(e, e->ac_tbl, nb << 4);
    if (nb) pe_bits(e, e->EOBRUN, nb);
    e->EOBRUN = 0;
    for (unsigned i = 0; i < e->BE; i++) pe_bits(e, e->corr[i], 1);
    e->BE = 0;
  }
}

static void pe_block(penc_t *e, const pscan_t *sc, int ci, const int16_t *blk, int *last_dc) {
  const int Al = sc->Al, tbl = ci ? 1 : 0;
  if (sc->Ss == 0) {
    if (sc->Ah == 0) {                      /* DC first: the point-transformed value, difference coded as in F.1.2.1 */
      int v = blk[0] >> Al;                 /* arithmetic shift */
      int t = v - last_dc[ci], t2 = t; last_dc[ci] = v;
      if (t < 0) { t = -t; t2--; }
      int n = nbits_of(t);
      pe_symbol(e, tbl, n);
      if (n) pe_bits(e, (unsigned)t2, n);
    } else {
      pe_bits(e, (unsigned)(blk[0] >> Al) & 1u, 1);
    }
    return;
  }
  if (sc->Ah == 0) {                        /* AC first (figure G.3) */
    int r = 0;
    for (int k = sc->Ss; k <= sc->Se; k++) {
      int t = blk[k], t2;
      if (t == 0) { r++; continue; }
      if (t < 0) { t = -t; t >>= Al; t2 = ~t; } else { t >>= Al; t2 = t; }
      if (t == 0) { r++; continue; }
      if (e->EOBRUN > 0) pe_eobrun(e);
      while (r > 15) { pe_symbol(e, tbl, 0xF0); r -= 16; }
      int n = nbits_of(t);
      pe_symbol(e, tbl, (r << 4) + n);
      pe_bits(e, (unsigned)t2, n);
      r = 0;
    }
    if (r > 0) { e->EOBRUN++; if (e->EOBRUN == 0x7FFF) pe_eobrun(e); }
    return;
  }
  /* AC refinement (figures G.5 - G.7) */
  int absv[64], EOB = 0;
  for (int k = sc->Ss; k <= sc->Se; k++) {
    int t = blk[k]; if (t < 0) t = -t;
    t >>= Al; absv[k] = t;
    if (t == 1) EOB = k;                    /* last coefficient that becomes non-zero in this scan */
  }
  int r = 0;
  unsigned BR = 0;
  uint8_t *BRbuf = e->corr + e->BE;         /* appended after the bits already buffered behind the EOB run */
  for (int k = sc->Ss; k <= sc->Se; k++) {
    int t = absv[k];
    if (t == 0) { r++; continue; }
    while (r > 15 && k <= EOB) {
      pe_eobrun(e);
      pe_symbol(e, tbl, 0xF0); r -= 16;
      for (unsigned i = 0; i < BR; i++) pe_bits(e, BRbuf[i], 1);
      BRbuf = e->corr; BR = 0;
    }
    if (t > 1) { BRbuf[BR++] = (uint8_t)(t & 1); continue; }     /* already non-zero: one correction bit */
    pe_eobrun(e);
    pe_symbol(e, tbl, (r << 4) + 1);
    pe_bits(e, blk[k] < 0 ? 0u : 1u, 1);
    for (unsigned i = 0; i < BR; i++) pe_bits(e, BRbuf[i], 1);
    BRbuf = e->corr; BR = 0; r = 0;
  }
  if (r > 0 || BR > 0) {
    e->EOBRUN++; e->BE += BR;
    if (e->EOBRUN == 0x7FFF || e->BE > (MJO_MAX_CORR_BITS - 64 + 1)) pe_eobrun(e);
  }
}

/* One pass over a scan: statistics when e->w == NULL, output otherwise. */
static void pe_scan(penc_t *e, const pscan_t *sc, const geom_t *g, int W, int H, const int16_t *coef, int restart_interval) {
  const int nl = g->hs * g->vs;
  int last_dc[3] = {0, 0, 0}, rst = 0;
  e->EOBRUN = 0; e->BE = 0;
  e->ac_tbl = sc->comp[0] ? 1 : 0;
  long nmcu; int bw = 0;
  if (sc->ncomp > 1) nmcu = (long)g->mcux * g->mcuy;
  else {
    const int c = sc->comp[0];
    const int cw = c == 0 ? W : ceil_div(W, g->hs), ch = c == 0 ? H : ceil_div(H, g->vs);
    bw = ceil_div(cw, 8);
    nmcu = (long)bw * ceil_div(ch, 8);
  }
  for (long m = 0; m < nmcu; m++) {
    if (restart_interval && m && m % restart_interval == 0) {
      pe_eobrun(e);
      if (e->w) { bw_flush(e->w); bw_byte(e->w, 0xFF); bw_byte(e->w, 0xD0 + rst); }
      rst = (rst + 1) & 7;
      last_dc[0] = last_dc[1] = last_dc[2] = 0;
      e->EOBRUN = 0; e->BE = 0;
    }
    if (sc->ncomp > 1) {
      for (int b = 0; b < g->bpm; b++) pe_block(e, sc, b < nl ? 0 : b - nl + 1, coef + ((size_t)m * g->bpm + b) * 64, last_dc);
    } else {
      const int c = sc->comp[0], bx = (int)(m % bw), by = (int)(m / bw);
      size_t slot = c == 0 ? ((size_t)(by / g->vs) * g->mcux + bx / g->hs) * g->bpm + (by % g->vs) * g->hs + bx % g->hs
                           : ((size_t)by * g->mcux + bx) * g->bpm + nl + c - 1;
      pe_block(e, sc, c, coef + slot * 64, last_dc);
    }
  }
  pe_eobrun(e);
  if (e->w) bw_flush(e->w);
}

MJO_API int mjo_encode_progressive_coefficients(const int16_t *coef, int W, int H, int quality, int css,
                                                int restart_interval, uint8_t **out, size_t *out_len) {
  geom_t g;
  if (make_geom(&g, W, H, css)) return -1;
  if (restart_interval < 0 || restart_interval > 65535) return -1;
  bw_t w; memset(&w, 0, sizeof w);
  uint16_t qt[2][64];
  mjo_quant_table(quality, 0, qt[0]); mjo_quant_table(quality, 1, qt[1]);
  bw_u16(&w, 0xFFD8);
  bw_u16(&w, 0xFFE0); bw_u16(&w, 16);
  bw_byte(&w, 'J'); bw_byte(&w, 'F'); bw_byte(&w, 'I'); bw_byte(&w, 'F'); bw_byte(&w, 0);
  bw_byte(&w, 1); bw_byte(&w, 1); bw_byte(&w, 0); bw_u16(&w, 1); bw_u16(&w, 1); bw_byte(&w, 0); bw_byte(&w, 0);
  for (int t = 0; t < 2; t++) {
    bw_u16(&w, 0xFFDB); bw_u16(&w, 67); bw_byte(&w, t);
    for (int k = 0; k < 64; k++) bw_byte(&w, qt[t][k_zigzag[k]]);
  }
  bw_u16(&w, 0xFFC2); bw_u16(&w, 17); bw_byte(&w, 8); bw_u16(&w, H); bw_u16(&w, W); bw_byte(&w, 3);
  bw_byte(&w, 1); bw_byte(&w, (g.hs << 4) | g.vs); bw_byte(&w, 0);
  bw_byte(&w, 2); bw_byte(&w, 0x11); bw_byte(&w, 1);
  bw_byte(&w, 3); bw_byte(&w, 0x11); bw_byte(&w, 1);
  uint32_t *cnt = (uint32_t *)malloc(2 * 257 * sizeof(uint32_t));
  if (!cnt) { free(w.p); return -2; }
  int dri_sent = 0;
  for (int s = 0; s < 10; s++) {
    const pscan_t *sc = &k_simple_progression[s];
    penc_t e; memset(&e, 0, sizeof e);
    htab_t tab[2];
    const int need_tab = !(sc->Ss == 0 && sc->Ah != 0);      /* a DC refinement scan has no Huffman-coded symbols */
    if (need_tab) {
      memset(cnt, 0, 2 * 257 * sizeof(uint32_t));
      e.w = NULL; e.count[0] = cnt; e.count[1] = cnt + 257;
      pe_scan(&e, sc, &g, W, H, coef, restart_interval);
      for (int t = 0; t < 2; t++) {
        int used = sc->ncomp > 1 ? 1 : ((sc->comp[0] ? 1 : 0) == t);
        if (!used) continue;
        memset(&tab[t], 0, sizeof tab[t]);
        if (mjo_gen_optimal_table(cnt + t * 257, tab[t].bits, tab[t].vals) < 0) { free(cnt); free(w.p); return -3; }
        derive_codes(&tab[t]);
        write_dht(&w, &tab[t], (sc->Ss == 0 ? 0x00 : 0x10) | t);
      }
    }
    if (restart_interval && !dri_sent) { bw_u16(&w, 0xFFDD); bw_u16(&w, 4); bw_u16(&w, restart_interval); dri_sent = 1; }
    bw_u16(&w, 0xFFDA); bw_u16(&w, 6 + 2 * sc->ncomp); bw_byte(&w, sc->ncomp);
    for (int i = 0; i < sc->ncomp; i++) {
      const int c = sc->comp[i], t = c ? 1 : 0;
      bw_byte(&w, c + 1);
      /* jcmarker.c emit_sos: a progressive scan names only the table kind it uses (DC xor AC); the other nibble is 0,
       * and a DC refinement scan names none */
      bw_byte(&w, sc->Ss == 0 ? (sc->Ah == 0 ? (t << 4) : 0) : t);
    }
    bw_byte(&w, sc->Ss); bw_byte(&w, sc->Se); bw_byte(&w, (sc->Ah << 4) | sc->Al);
    e.w = &w; e.tab[0] = &tab[0]; e.tab[1] = &tab[1];
    pe_scan(&e, sc, &g, W, H, coef, restart_interval);
  }
  free(cnt);
  bw_u16(&w, 0xFFD9);
  *out = w.p; *out_len = w.n;
  return 0;
}

MJO_API int mjo_encode_progressive(const uint8_t *src, int W, int H, size_t stride, int pixfmt, int quality, int css,
                                   int restart_interval, uint8_t **out, size_t *out_len) {
  geom_t g;
  if (make_geom(&g, W, H, css)) return -1;
  size_t n = (size_t)g.mcux * g.mcuy * g.bpm * 64;
  int16_t *coef = (int16_t *)malloc(n * sizeof(int16_t));
  if (!coef) return -2;
  int rc = mjo_coefficients(src, W, H, stride, pixfmt, quality, css, coef);
  if (!rc) rc = mjo_encode_progressive_coefficients(coef, W, H, quality, css, restart_interval, out, out_len);
  free(coef);
  return rc;
}

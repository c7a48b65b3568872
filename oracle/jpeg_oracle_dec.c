/*
 * jpeg_oracle_dec.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * CPU restatement of the JPEG decode path that the reference delegates to nvJPEG:
 *   reference call sites  src/ImageCompressorDll/ImageCompressorImpl.cu:335 (nvjpegGetImageInfo),
 *                         :361-366 (JpegStreamParse, DecodeJpegHost, TransferToDevice, DecodeJpegDevice),
 *                         :184-232 (getCVImageOnCPU: planar B,G,R -> interleaved BGR cv::Mat)
 * PARITY STATUS: parity unpinned against nvJPEG itself (closed source, no fixtures in the reference). Pinned, byte for
 * byte, against the stock decoder in this image -- libjpeg-turbo 3.1.4.1 with its defaults (what Pillow uses):
 * "islow" accurate integer IDCT, fancy (triangle) chroma upsampling for 2x factors, pixel replication otherwise,
 * 16.16 fixed-point YCbCr->RGB -- by tests/test_oracle_dec_pin.py.
 *
 * Handles baseline sequential (SOF0) 3-component YCbCr files with chroma 1x1 and luma h x v in {1,2,4} x {1,2}
 * (everything this project's encoder writes, with or without restart intervals), Huffman tables from the file.
 *
 * ATTRIBUTION: the inverse DCT, the fancy upsampling filters and the YCbCr->RGB tables RESTATE the integer procedures of
 * the Independent JPEG Group's libjpeg / libjpeg-turbo (jidctint.c, jdsample.c, jdcolor.c, jdhuff.c), because pixel
 * identity with the stock decoder requires exactly their arithmetic.
 *   This software is based in part on the work of the Independent JPEG Group.
 *   libjpeg: Copyright (C) 1991-2020, Thomas G. Lane, Guido Vollbeding. libjpeg-turbo: Copyright (C) 2009-2024
 *   D. R. Commander et al., IJG licence + Modified (3-clause) BSD licence (libjpeg-turbo LICENSE.md / README.ijg).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MJO_API __attribute__((visibility("default")))

static const uint8_t k_zz[64] = {
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
  28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
  47, 55, 62, 63};

typedef struct {
  int W, H, hs, vs, ri;
  uint16_t q[2][64];          /* natural order */
  uint8_t bits[4][17], vals[4][256];   /* [DC0, AC0, DC1, AC1] */
  int have_tab[4];
  int tq[3], td[3], ta[3];
  const uint8_t *scan; size_t scan_len;
} hdr_t;

static int parse(const uint8_t *p, size_t n, hdr_t *h) {
  memset(h, 0, sizeof *h);
  if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return -1;
  size_t i = 2;
  while (i + 4 <= n) {
    if (p[i] != 0xFF) return -2;
    int m = p[i + 1];
    if (m == 0xFF) { i++; continue; }
    size_t len = ((size_t)p[i + 2] << 8) | p[i + 3];
    if (i + 2 + len > n) return -3;
    const uint8_t *s = p + i + 4;
    if (m == 0xDB) {
      size_t k = 0;
      while (k + 65 <= len - 2) {
        int pq = s[k] >> 4, t = s[k] & 15;
        if (pq || t > 1) return -4;
        for (int z = 0; z < 64; z++) h->q[t][k_zz[z]] = s[k + 1 + z];
        k += 65;
      }
    } else if (m == 0xC0) {
      if (s[0] != 8 || s[5] != 3) return -5;
      h->H = (s[1] << 8) | s[2]; h->W = (s[3] << 8) | s[4];
      for (int c = 0; c < 3; c++) {
        int hv = s[7 + 3 * c];
        h->tq[c] = s[8 + 3 * c];
        if (c == 0) { h->hs = hv >> 4; h->vs = hv & 15; } else if (hv != 0x11) return -6;
      }
    } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      return -7;   /* progressive, lossless, arithmetic: not handled */
    } else if (m == 0xC4) {
      size_t k = 0;
      while (k + 17 <= len - 2) {
        int tc = s[k] >> 4, th = s[k] & 15;
        if (tc > 1 || th > 1) return -8;
        int t = th * 2 + tc, cnt = 0;
        h->bits[t][0] = 0;
        for (int l = 1; l <= 16; l++) { h->bits[t][l] = s[k + l]; cnt += s[k + l]; }
        if (cnt > 256 || k + 17 + cnt > len - 2) return -9;
        memcpy(h->vals[t], s + k + 17, (size_t)cnt);
        h->have_tab[t] = 1;
        k += 17 + (size_t)cnt;
      }
    } else if (m == 0xDD) {
      h->ri = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (s[0] != 3) return -10;
      for (int c = 0; c < 3; c++) { h->td[c] = s[2 + 2 * c] >> 4; h->ta[c] = s[2 + 2 * c] & 15; }
      h->scan = p + i + 2 + len;
      h->scan_len = n - (i + 2 + len);
      return (h->W > 0 && h->H > 0 && h->hs >= 1 && h->vs >= 1) ? 0 : -11;
    }
    i += 2 + len;
  }
  return -12;
}

/* info[0..4] = width, height, luma h, luma v, restart interval */
MJO_API int mjo_decode_info(const uint8_t *jpg, size_t n, int32_t *info) {
  hdr_t h;
  int rc = parse(jpg, n, &h);
  if (rc) return rc;
  info[0] = h.W; info[1] = h.H; info[2] = h.hs; info[3] = h.vs; info[4] = h.ri;
  return 0;
}

/* ---- Huffman decoding (T.81 F.2.2) ---- */
typedef struct { int mincode[17], maxcode[18], valptr[17]; const uint8_t *vals; } dtab_t;
static void build_dtab(dtab_t *d, const uint8_t *bits, const uint8_t *vals) {
  int code = 0, p = 0;
  for (int l = 1; l <= 16; l++) {
    d->valptr[l] = p; d->mincode[l] = code;
    code += bits[l]; p += bits[l];
    d->maxcode[l] = bits[l] ? code - 1 : -1;
    code <<= 1;
  }
  d->maxcode[17] = 0x7FFFFFFF;
  d->vals = vals;
}
typedef struct { const uint8_t *p, *end; uint32_t acc; int n; int hit_marker; } br_t;
static inline int br_bit(br_t *b) {
  if (b->n == 0) {
    int v = 0;
    if (b->p < b->end && !b->hit_marker) {
      v = *b->p;
      if (v == 0xFF) {
        if (b->p + 1 < b->end && b->p[1] == 0) b->p += 2;
        else { b->hit_marker = 1; v = 0; }      /* marker: feed zeros */
      } else b->p++;
    }
    b->acc = (uint32_t)v; b->n = 8;
  }
  b->n--;
  return (b->acc >> b->n) & 1;
}
static inline int br_bits(br_t *b, int k) { int v = 0; while (k--) v = (v << 1) | br_bit(b); return v; }
static inline int huff(br_t *b, const dtab_t *d) {
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br_bit(b);
    if (d->maxcode[l] >= 0 && code <= d->maxcode[l] && code >= d->mincode[l]) return d->vals[d->valptr[l] + code - d->mincode[l]];
  }
  return 0;
}
static inline int extend(int v, int s) { return s && v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

/* Entropy-decode the scan into quantised coefficients, MCU order, zig-zag order inside a block (the same layout the
 * encoder side uses). coef must hold mcux*mcuy*(hs*vs+2)*64 int16. */
static int decode_coefficients(const hdr_t *h, int16_t *coef) {
  const int mcux = (h->W + 8 * h->hs - 1) / (8 * h->hs), mcuy = (h->H + 8 * h->vs - 1) / (8 * h->vs);
  const int nl = h->hs * h->vs, bpm = nl + 2;
  dtab_t t[4];
  for (int i = 0; i < 4; i++) { if (!h->have_tab[i]) return -20; build_dtab(&t[i], h->bits[i], h->vals[i]); }
  br_t b = {h->scan, h->scan + h->scan_len, 0, 0, 0};
  int pred[3] = {0, 0, 0};
  long nmcu = (long)mcux * mcuy;
  memset(coef, 0, (size_t)nmcu * bpm * 64 * sizeof(int16_t));
  for (long m = 0; m < nmcu; m++) {
    if (h->ri && m && m % h->ri == 0) {
      /* byte-align, expect RSTn */
      b.n = 0;
      if (b.hit_marker) b.hit_marker = 0;
      while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
      if (b.p + 1 < b.end) b.p += 2;
      pred[0] = pred[1] = pred[2] = 0;
    }
    for (int k = 0; k < bpm; k++) {
      const int c = k < nl ? 0 : k - nl + 1;
      int16_t *blk = coef + ((size_t)m * bpm + k) * 64;
      const dtab_t *dc = &t[h->td[c] * 2], *ac = &t[h->ta[c] * 2 + 1];
      int s = huff(&b, dc);
      int diff = extend(br_bits(&b, s), s);
      pred[c] += diff;
      blk[0] = (int16_t)pred[c];
      for (int z = 1; z < 64;) {
        int rs = huff(&b, ac), r = rs >> 4, sz = rs & 15;
        if (sz == 0) { if (r == 15) { z += 16; continue; } break; }
        z += r;
        if (z > 63) return -21;
        blk[z] = (int16_t)extend(br_bits(&b, sz), sz);
        z++;
      }
    }
  }
  return 0;
}

/* ---- inverse DCT: accurate integer ("islow"), 13-bit constants, dequantisation folded in ---- */
#define CB 13
#define P1 2
#define DS(x, n) (((x) + (1 << ((n)-1))) >> (n))
static inline uint8_t clamp8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
static void idct_islow(const int16_t *zzcoef, const uint16_t *q, uint8_t *out, int stride) {
  int ws[64], in[64];
  for (int z = 0; z < 64; z++) in[k_zz[z]] = zzcoef[z] * q[k_zz[z]];
  for (int c = 0; c < 8; c++) {
    const int *p = in + c; int *w = ws + c;
    if (!(p[8] | p[16] | p[24] | p[32] | p[40] | p[48] | p[56])) {
      int dc = p[0] << P1;
      for (int r = 0; r < 8; r++) w[8 * r] = dc;
      continue;
    }
    int z2 = p[16], z3 = p[48];
    int z1 = (z2 + z3) * 4433;
    int t2 = z1 + z3 * (-15137), t3 = z1 + z2 * 6270;
    z2 = p[0]; z3 = p[32];
    int t0 = (z2 + z3) << CB, t1 = (z2 - z3) << CB;
    int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = p[56]; t1 = p[40]; t2 = p[24]; t3 = p[8];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; int z4 = t1 + t3;
    int z5 = (z3 + z4) * 9633;
    t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    w[0] = DS(t10 + t3, CB - P1); w[56] = DS(t10 - t3, CB - P1);
    w[8] = DS(t11 + t2, CB - P1); w[48] = DS(t11 - t2, CB - P1);
    w[16] = DS(t12 + t1, CB - P1); w[40] = DS(t12 - t1, CB - P1);
    w[24] = DS(t13 + t0, CB - P1); w[32] = DS(t13 - t0, CB - P1);
  }
  for (int r = 0; r < 8; r++) {
    const int *w = ws + 8 * r; uint8_t *o = out + (size_t)r * stride;
    int z2 = w[2], z3 = w[6];
    int z1 = (z2 + z3) * 4433;
    int t2 = z1 + z3 * (-15137), t3 = z1 + z2 * 6270;
    int t0 = (w[0] + w[4]) << CB, t1 = (w[0] - w[4]) << CB;
    int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    t0 = w[7]; t1 = w[5]; t2 = w[3]; t3 = w[1];
    z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2; int z4 = t1 + t3;
    int z5 = (z3 + z4) * 9633;
    t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
    const int SH = CB + P1 + 3;
    o[0] = clamp8(DS(t10 + t3, SH) + 128); o[7] = clamp8(DS(t10 - t3, SH) + 128);
    o[1] = clamp8(DS(t11 + t2, SH) + 128); o[6] = clamp8(DS(t11 - t2, SH) + 128);
    o[2] = clamp8(DS(t12 + t1, SH) + 128); o[5] = clamp8(DS(t12 - t1, SH) + 128);
    o[3] = clamp8(DS(t13 + t0, SH) + 128); o[4] = clamp8(DS(t13 - t0, SH) + 128);
  }
}

/* ---- chroma upsampling: triangle ("fancy") filters for h2v1 / h1v2 / h2v2, replication otherwise ---- */
static void upsample(const uint8_t *c, int cw, int ch, int cstride, int hs, int vs, uint8_t *out, int W, int H) {
  /* c: chroma plane, cw x ch valid samples (cw = ceil(W/hs), ch = ceil(H/vs)); out: W x H */
  if (hs == 1 && vs == 1) {
    for (int y = 0; y < H; y++) memcpy(out + (size_t)y * W, c + (size_t)y * cstride, (size_t)W);
  } else if (hs == 2 && vs == 1 && cw > 2) {   /* the stock decoder uses the triangle filter only when the plane is > 2 wide */
    for (int y = 0; y < H; y++) {
      const uint8_t *s = c + (size_t)y * cstride; uint8_t *o = out + (size_t)y * W;
      for (int x = 0; x < W; x++) {
        int i = x >> 1, v = s[i];
        if (x & 1) o[x] = (uint8_t)(i + 1 < cw ? (3 * v + s[i + 1] + 2) >> 2 : v);
        else o[x] = (uint8_t)(i > 0 ? (3 * v + s[i - 1] + 1) >> 2 : v);
      }
    }
  } else if (hs == 1 && vs == 2) {
    for (int y = 0; y < H; y++) {
      int i = y >> 1;
      const uint8_t *s0 = c + (size_t)i * cstride;
      const uint8_t *s1 = (y & 1) ? (i + 1 < ch ? s0 + cstride : s0) : (i > 0 ? s0 - cstride : s0);
      int bias = (y & 1) ? 2 : 1;
      uint8_t *o = out + (size_t)y * W;
      for (int x = 0; x < W; x++) o[x] = (uint8_t)((3 * s0[x] + s1[x] + bias) >> 2);
    }
  } else if (hs == 2 && vs == 2 && cw > 2) {
    for (int y = 0; y < H; y++) {
      int i = y >> 1;
      const uint8_t *s0 = c + (size_t)i * cstride;
      const uint8_t *s1 = (y & 1) ? (i + 1 < ch ? s0 + cstride : s0) : (i > 0 ? s0 - cstride : s0);
      uint8_t *o = out + (size_t)y * W;
      for (int x = 0; x < W; x++) {
        int j = x >> 1;
        int cur = 3 * s0[j] + s1[j];
        if (x & 1) { int nx = j + 1 < cw ? 3 * s0[j + 1] + s1[j + 1] : -1; o[x] = (uint8_t)(nx >= 0 ? (cur * 3 + nx + 7) >> 4 : (cur * 4 + 7) >> 4); }
        else { int pv = j > 0 ? 3 * s0[j - 1] + s1[j - 1] : -1; o[x] = (uint8_t)(pv >= 0 ? (cur * 3 + pv + 8) >> 4 : (cur * 4 + 8) >> 4); }
      }
    }
  } else {
    for (int y = 0; y < H; y++) {
      const uint8_t *s = c + (size_t)(y / vs) * cstride; uint8_t *o = out + (size_t)y * W;
      for (int x = 0; x < W; x++) o[x] = s[x / hs];
    }
  }
}

#define FIXD(x) ((int32_t)((x) * 65536.0 + 0.5))
/* out pixfmt: 0 = RGB interleaved, 1 = BGR interleaved (cv::Mat CV_8UC3, reference getCVImageOnCPU .cu:214-221) */
MJO_API int mjo_decode(const uint8_t *jpg, size_t n, int pixfmt, uint8_t *out, size_t stride, int *w_out, int *h_out) {
  hdr_t h;
  int rc = parse(jpg, n, &h);
  if (rc) return rc;
  if (!((h.hs == 1 || h.hs == 2 || h.hs == 4) && (h.vs == 1 || h.vs == 2))) return -30;
  const int W = h.W, H = h.H, hs = h.hs, vs = h.vs;
  const int mcux = (W + 8 * hs - 1) / (8 * hs), mcuy = (H + 8 * vs - 1) / (8 * vs), nl = hs * vs, bpm = nl + 2;
  int16_t *coef = (int16_t *)malloc((size_t)mcux * mcuy * bpm * 64 * sizeof(int16_t));
  if (!coef) return -31;
  rc = decode_coefficients(&h, coef);
  if (rc) { free(coef); return rc; }
  const int yw = mcux * hs * 8, yh = mcuy * vs * 8, cwp = mcux * 8, chp = mcuy * 8;
  uint8_t *Y = (uint8_t *)malloc((size_t)yw * yh), *C[2] = {(uint8_t *)malloc((size_t)cwp * chp), (uint8_t *)malloc((size_t)cwp * chp)};
  uint8_t *U[2] = {(uint8_t *)malloc((size_t)W * H), (uint8_t *)malloc((size_t)W * H)};
  for (int my = 0; my < mcuy; my++)
    for (int mx = 0; mx < mcux; mx++) {
      const int16_t *m = coef + ((size_t)my * mcux + mx) * bpm * 64;
      for (int yi = 0; yi < vs; yi++)
        for (int xi = 0; xi < hs; xi++)
          idct_islow(m + (yi * hs + xi) * 64, h.q[h.tq[0]], Y + (size_t)((my * vs + yi) * 8) * yw + (mx * hs + xi) * 8, yw);
      for (int c = 0; c < 2; c++) idct_islow(m + (nl + c) * 64, h.q[h.tq[1 + c]], C[c] + (size_t)(my * 8) * cwp + mx * 8, cwp);
    }
  const int cw = (W + hs - 1) / hs, ch = (H + vs - 1) / vs;
  for (int c = 0; c < 2; c++) upsample(C[c], cw, ch, cwp, hs, vs, U[c], W, H);
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) {
      int yy = Y[(size_t)y * yw + x], cb = U[0][(size_t)y * W + x] - 128, cr = U[1][(size_t)y * W + x] - 128;
      int r = yy + ((FIXD(1.40200) * cr + 32768) >> 16);
      int g = yy + ((-FIXD(0.34414) * cb + 32768 - FIXD(0.71414) * cr) >> 16);
      int b = yy + ((FIXD(1.77200) * cb + 32768) >> 16);
      uint8_t *o = out + (size_t)y * stride + (size_t)x * 3;
      if (pixfmt == 0) { o[0] = clamp8(r); o[1] = clamp8(g); o[2] = clamp8(b); } else { o[0] = clamp8(b); o[1] = clamp8(g); o[2] = clamp8(r); }
    }
  if (w_out) *w_out = W;
  if (h_out) *h_out = H;
  free(coef); free(Y); free(C[0]); free(C[1]); free(U[0]); free(U[1]);
  return 0;
}

/* Quantised coefficients of a file, in the encoder's layout (used to check the entropy decoder on its own). */
MJO_API int mjo_decode_coefficients(const uint8_t *jpg, size_t n, int16_t *coef) {
  hdr_t h;
  int rc = parse(jpg, n, &h);
  if (rc) return rc;
  return decode_coefficients(&h, coef);
}

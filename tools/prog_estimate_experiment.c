// Feasibility experiment (not product code): can the start of a block of an AC REFINEMENT scan be located by hypothesis
// elimination? Full progressive entropy decode of a 3-component file without restart markers (T.81 G.1.2, jdphuff.c's procedures),
// recording for every AC refinement scan the bit position at which each block starts and the non-zero history mask of each block;
// then, for sampled anchor blocks, every bit position in a window around the true start is tried as a hypothesis and parsed
// forward (masks of the following blocks, no coefficient writes) until a violation shows or N blocks have passed.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>

typedef struct { int look[65536]; } HT;   // (len << 8) | sym, 0 = invalid
static HT *ht_dc[4], *ht_ac[4];
static void build_ht(HT *t, const uint8_t *bits, const uint8_t *vals) {
  memset(t, 0, sizeof *t);
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < bits[l]; i++, k++, code++) {
      int lo = code << (16 - l), n = 1 << (16 - l);
      for (int j = 0; j < n; j++) t->look[lo + j] = (l << 8) | vals[k];
    }
    code <<= 1;
  }
}
typedef struct { const uint8_t *d; size_t nbits; size_t pos; int bad; } BR;   // unstuffed data, bit position
static inline uint32_t peek16(BR *b) {
  size_t byte = b->pos >> 3; int off = b->pos & 7;
  uint32_t w = ((uint32_t)b->d[byte] << 16) | ((uint32_t)b->d[byte + 1] << 8) | b->d[byte + 2];
  return (w >> (8 - off)) & 0xFFFF;
}
static inline uint32_t getbits(BR *b, int s) {
  if (!s) return 0;
  uint32_t v = peek16(b) >> (16 - s);
  b->pos += s;
  if (b->pos > b->nbits) b->bad = 1;
  return v;
}
static inline int decode(BR *b, HT *t) {
  int e = t->look[peek16(b)];
  if (!e) { b->bad = 1; b->pos += 16; return 0; }
  b->pos += e >> 8;
  if (b->pos > b->nbits) b->bad = 1;
  return e & 255;
}
static int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

static int W, H, ncomp, hs[3], vs[3], hmax, vmax;
static int bw[3], bh[3];          // component size in blocks (non-interleaved scan geometry)
static int mcux, mcuy;
static int16_t *coef[3];          // [blocks (padded grid)][64] zig-zag order
static int pw[3];                 // padded width in blocks
static float *firstbits[3];       // bits the AC first scans spent on each block (by block index of the non-interleaved scan order)
typedef struct { int ncomp, comp[3], td[3], ta[3], Ss, Se, Ah, Al; uint8_t *data; size_t nbytes; } Scan;

// parse one refinement block from (b->pos) with history mask `hist` (bit k = coefficient k already non-zero); returns 0 ok, 1 violation.
// *eobrun in/out. No coefficient writes.
static int parse_refine_block(BR *br, HT *t, uint64_t hist, int Ss, int Se, int *eobrun) {
  int k = Ss;
  if (*eobrun == 0) {
    for (; k <= Se; k++) {
      int rs = decode(br, t);
      if (br->bad) return 1;
      int r = rs >> 4, s = rs & 15;
      if (s) { if (s != 1) return 1; getbits(br, 1); }
      else if (r != 15) { *eobrun = (1 << r) + (int)getbits(br, r); break; }
      int zr = r;
      do {
        if ((hist >> k) & 1) getbits(br, 1);
        else if (--zr < 0) break;
        k++;
      } while (k <= Se);
      if (k > Se) return 1;       // ran out of band before the run ended: a valid stream never does (for s != 0 and for ZRL alike)
      if (br->bad) return 1;
    }
  }
  if (*eobrun > 0) {
    for (; k <= Se; k++) if ((hist >> k) & 1) getbits(br, 1);
    (*eobrun)--;
  }
  return br->bad;
}

int main(int argc, char **argv) {
  FILE *f = fopen(argv[1], "rb");
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  uint8_t *j = malloc(n + 16); fread(j, 1, n, f); fclose(f);
  int window = argc > 2 ? atoi(argv[2]) : 2000, nanch = argc > 3 ? atoi(argv[3]) : 40, maxblocks = argc > 4 ? atoi(argv[4]) : 200;
  uint8_t dcb[4][17], dcv[4][256], acb[4][17], acv[4][256];
  for (int i = 0; i < 4; i++) { ht_dc[i] = malloc(sizeof(HT)); ht_ac[i] = malloc(sizeof(HT)); }
  long p = 2;
  int scan_no = 0;
  while (p < n) {
    if (j[p] != 0xFF) { p++; continue; }
    int m = j[p + 1];
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7) || m == 0xFF) { p += (m == 0xFF) ? 1 : 2; continue; }
    if (m == 0xD9) break;
    int L = (j[p + 2] << 8) | j[p + 3];
    uint8_t *q = j + p + 4;
    if (m == 0xC2 || m == 0xC0 || m == 0xC1) {
      H = (q[1] << 8) | q[2]; W = (q[3] << 8) | q[4]; ncomp = q[5];
      hmax = vmax = 1;
      for (int c = 0; c < ncomp; c++) { hs[c] = q[7 + 3 * c] >> 4; vs[c] = q[7 + 3 * c] & 15; if (hs[c] > hmax) hmax = hs[c]; if (vs[c] > vmax) vmax = vs[c]; }
      mcux = (W + 8 * hmax - 1) / (8 * hmax); mcuy = (H + 8 * vmax - 1) / (8 * vmax);
      for (int c = 0; c < ncomp; c++) {
        int cw = (W * hs[c] + hmax - 1) / hmax, ch = (H * vs[c] + vmax - 1) / vmax;
        bw[c] = (cw + 7) / 8; bh[c] = (ch + 7) / 8; pw[c] = mcux * hs[c];
        coef[c] = calloc((size_t)pw[c] * mcuy * vs[c] * 64, 2); firstbits[c] = calloc((size_t)pw[c] * mcuy * vs[c] + 16, 4);
      }
      fprintf(stderr, "%dx%d comps %d, luma %dx%d, mcus %dx%d\n", W, H, ncomp, hs[0], vs[0], mcux, mcuy);
    } else if (m == 0xC4) {
      uint8_t *e = q + L - 2;
      while (q < e) {
        int tc = q[0] >> 4, th = q[0] & 15, cnt = 0;
        uint8_t *bits = tc ? acb[th] : dcb[th], *vals = tc ? acv[th] : dcv[th];
        bits[0] = 0;
        for (int i = 1; i <= 16; i++) { bits[i] = q[i]; cnt += q[i]; }
        memcpy(vals, q + 17, cnt);
        build_ht(tc ? ht_ac[th] : ht_dc[th], bits, vals);
        q += 17 + cnt;
      }
    } else if (m == 0xDA) {
      Scan s; s.ncomp = q[0];
      for (int i = 0; i < s.ncomp; i++) { s.comp[i] = q[1 + 2 * i] - 1; s.td[i] = q[2 + 2 * i] >> 4; s.ta[i] = q[2 + 2 * i] & 15; }
      s.Ss = q[1 + 2 * s.ncomp]; s.Se = q[2 + 2 * s.ncomp]; s.Ah = q[3 + 2 * s.ncomp] >> 4; s.Al = q[3 + 2 * s.ncomp] & 15;
      long d0 = p + 2 + L, d = d0;
      while (!(j[d] == 0xFF && j[d + 1] != 0 && !(j[d + 1] >= 0xD0 && j[d + 1] <= 0xD7))) d++;
      s.data = malloc(d - d0 + 16); s.nbytes = 0;
      for (long i = d0; i < d; i++) { s.data[s.nbytes++] = j[i]; if (j[i] == 0xFF) i++; }
      memset(s.data + s.nbytes, 0, 16);
      BR br = {s.data, s.nbytes * 8, 0, 0};
      int p1 = 1 << s.Al, m1 = -(1 << s.Al);
      fprintf(stderr, "scan %d: comps %d (first %d) Ss %d Se %d Ah %d Al %d, %zu bytes\n", scan_no, s.ncomp, s.comp[0], s.Ss, s.Se, s.Ah, s.Al, s.nbytes);
      if (s.Ss == 0) {
        // DC scans
        int pred[3] = {0, 0, 0};
        for (int my = 0; my < mcuy; my++) for (int mx = 0; mx < mcux; mx++)
          for (int ci = 0; ci < s.ncomp; ci++) { int c = s.comp[ci];
            for (int y = 0; y < vs[c]; y++) for (int x = 0; x < hs[c]; x++) {
              int16_t *b = coef[c] + ((size_t)(my * vs[c] + y) * pw[c] + mx * hs[c] + x) * 64;
              if (s.Ah == 0) { int t = decode(&br, ht_dc[s.td[ci]]); pred[ci] += t ? extend(getbits(&br, t), t) : 0; b[0] = pred[ci] * p1; }
              else if (getbits(&br, 1)) b[0] |= p1;
            } }
      } else {
        int c = s.comp[0];
        long nblk = (long)bw[c] * bh[c];
        size_t *startpos = NULL; uint64_t *hist = NULL;
        if (s.Ah) { startpos = malloc((nblk + 1) * sizeof(size_t)); hist = malloc(nblk * 8); }
        int eobrun = 0;
        long bi = 0, nsym = 0;
        for (int by = 0; by < bh[c]; by++) for (int bx = 0; bx < bw[c]; bx++, bi++) {
          int16_t *b = coef[c] + ((size_t)by * pw[c] + bx) * 64;
          if (s.Ah == 0) {
            size_t fb0 = br.pos;
            if (eobrun > 0) { eobrun--; continue; }
            for (int k = s.Ss; k <= s.Se; k++) {
              int rs = decode(&br, ht_ac[s.ta[0]]); int r = rs >> 4, t = rs & 15;
              if (t) { k += r; b[k] = extend(getbits(&br, t), t) * p1; }
              else if (r == 15) k += 15;
              else { eobrun = (1 << r) + getbits(&br, r) - 1; break; }
            }
            firstbits[c][bi] += (float)(br.pos - fb0);
          } else {
            uint64_t hm = 0;
            for (int k = s.Ss; k <= s.Se; k++) if (b[k]) hm |= 1ull << k;
            hist[bi] = hm; startpos[bi] = br.pos;
            int k = s.Ss;
            if (eobrun == 0) {
              for (; k <= s.Se; k++) {
                int rs = decode(&br, ht_ac[s.ta[0]]); int r = rs >> 4, t = rs & 15; nsym++;
                if (t) { t = getbits(&br, 1) ? p1 : m1; }
                else if (r != 15) { eobrun = (1 << r) + getbits(&br, r); break; }
                do { int v = b[k]; if (v) { if (getbits(&br, 1) && !(v & p1)) b[k] = v + (v >= 0 ? p1 : m1); } else if (--r < 0) break; k++; } while (k <= s.Se);
                if (t) b[k] = t;
              }
            }
            if (eobrun > 0) { for (; k <= s.Se; k++) { int v = b[k]; if (v && getbits(&br, 1) && !(v & p1)) b[k] = v + (v >= 0 ? p1 : m1); } eobrun--; }
          }
        }
        if (br.bad) fprintf(stderr, "  decode error!\n");
        if (s.Ah) {
          startpos[nblk] = br.pos;
          double avgbits = (double)br.pos / nblk; long dense = 0; for (long i = 0; i < nblk; i++) dense += __builtin_popcountll(hist[i]);
          fprintf(stderr, "  refinement: %ld blocks, %.1f bits/block, %.1f history coefs/block, %.1f symbols/block\n", nblk, avgbits, (double)dense / nblk, (double)nsym / nblk);
          {
            const int G = 2048; long nseg = nblk / G;
            double *y = malloc(nseg * 8), *x1 = malloc(nseg * 8), *x2 = malloc(nseg * 8), *x3 = malloc(nseg * 8);
            for (long i = 0; i < nseg; i++) {
              y[i] = (double)startpos[(i + 1) * G] - (double)startpos[i * G];
              double h = 0, fb = 0, nz1 = 0;
              for (long b2 = i * G; b2 < (i + 1) * G; b2++) { h += __builtin_popcountll(hist[b2]); fb += firstbits[c][b2]; 
                int by2 = b2 / bw[c], bx2 = b2 % bw[c]; int16_t *bb = coef[c] + ((size_t)by2 * pw[c] + bx2) * 64; (void)bb; }
              x1[i] = h; x2[i] = fb; x3[i] = nz1;
            }
            // models: (0) today's: bits ~ k (h + 8 G) with k from the two ends (bridge) -- evaluated as left-anchored with global k; (1) a + b h; (2) a + b h + c fb
            for (int model = 0; model < 3; model++) {
              double A[3][3] = {{0}}, B[3] = {0}, w[3] = {0};
              int nv = model == 0 ? 1 : model == 1 ? 2 : 3;
              for (long i = 0; i < nseg; i++) {
                double v[3]; if (model == 0) { v[0] = x1[i] + 8.0 * G; } else { v[0] = 1; v[1] = x1[i]; v[2] = x2[i]; }
                for (int r = 0; r < nv; r++) { B[r] += v[r] * y[i]; for (int q2 = 0; q2 < nv; q2++) A[r][q2] += v[r] * v[q2]; }
              }
              // solve nv x nv
              for (int r = 0; r < nv; r++) { double pv = A[r][r]; for (int q2 = r; q2 < nv; q2++) A[r][q2] /= pv; B[r] /= pv;
                for (int r2 = 0; r2 < nv; r2++) if (r2 != r) { double f2 = A[r2][r]; for (int q2 = r; q2 < nv; q2++) A[r2][q2] -= f2 * A[r][q2]; B[r2] -= f2 * B[r]; } }
              for (int r = 0; r < nv; r++) w[r] = B[r];
              double res2 = 0, ybar = 0; for (long i = 0; i < nseg; i++) ybar += y[i]; ybar /= nseg;
              double *res = malloc(nseg * 8);
              for (long i = 0; i < nseg; i++) { double v[3]; if (model == 0) { v[0] = x1[i] + 8.0 * G; } else { v[0] = 1; v[1] = x1[i]; v[2] = x2[i]; }
                double p2 = 0; for (int r = 0; r < nv; r++) p2 += w[r] * v[r]; res[i] = y[i] - p2; res2 += res[i] * res[i]; }
              fprintf(stderr, "    model %d: coefficients %.4f %.4f %.4f; per-segment residual rms %.0f bits (mean segment %.0f bits, sqrt %.0f)", model, w[0], w[1], w[2], sqrt(res2 / nseg), ybar, sqrt(ybar));
              // left-anchored drift over d segments
              for (int d = 1; d <= 64; d *= 4) { double s2 = 0; long n = 0; for (long i = 0; i + d <= nseg; i += d) { double acc = 0; for (int u = 0; u < d; u++) acc += res[i + u]; s2 += acc * acc; n++; }
                fprintf(stderr, " | drift over %d seg: rms %.0f", d, sqrt(s2 / (n ? n : 1))); }
              fprintf(stderr, "\n");
              free(res);
            }
            free(y); free(x1); free(x2); free(x3);
          }
        }
        free(startpos); free(hist);
      }
      free(s.data);
      scan_no++;
      p = d; continue;
    }
    p += 2 + L;
  }
  return 0;
}

/*
 * ijg_harness.c -- TEST INFRASTRUCTURE ONLY.
 * Thin command-line driver around the *stock* IJG libjpeg 9d that ships in this image
 * (/opt/conda/include/jpeglib.h, /opt/conda/lib/libjpeg.so.9). It is an independent JPEG code base used to pin
 * oracle/jpeg_oracle.c for the samplings Pillow's libjpeg-turbo build cannot produce (4:4:0, 4:1:1) and to
 * read back quantised coefficients / decode files produced by the oracle and by the HIP path.
 *
 *   ijg_harness enc  in.raw W H rgb|ycc quality hs vs optimize restart_mcus out.jpg
 *   ijg_harness coef in.jpg out.bin      (int16; per component: blocks in raster order, natural coefficient order;
 *                                         preceded by a header of int32: ncomp, then per comp wib, hib, hs, vs)
 *   ijg_harness dec  in.jpg out.raw      (RGB8 interleaved, library defaults)
 *   ijg_harness bench in.raw W H quality hs vs optimize restart_mcus reps
 *                                        (CPU baseline for 4:4:0 / 4:1:1, bench.py: encodes to memory `reps` times, prints
 *                                         "<best seconds> <bytes>"; file reading is outside the timed region)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <jpeglib.h>

static unsigned char *slurp(const char *path, size_t *n) {
  FILE *f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END); long sz = ftell(f); rewind(f);
  unsigned char *b = (unsigned char *)malloc((size_t)sz + 1);
  if (fread(b, 1, (size_t)sz, f) != (size_t)sz) { perror("read"); exit(2); }
  fclose(f); *n = (size_t)sz; return b;
}

static int do_enc(int argc, char **argv) {
  if (argc != 12) return 1;
  size_t n; unsigned char *raw = slurp(argv[2], &n);
  int W = atoi(argv[3]), H = atoi(argv[4]);
  int ycc = !strcmp(argv[5], "ycc");
  int quality = atoi(argv[6]), hs = atoi(argv[7]), vs = atoi(argv[8]), opt = atoi(argv[9]), rst = atoi(argv[10]);
  if (n < (size_t)W * H * 3) { fprintf(stderr, "short input\n"); return 2; }
  struct jpeg_compress_struct c; struct jpeg_error_mgr e;
  c.err = jpeg_std_error(&e);
  jpeg_create_compress(&c);
  FILE *fo = fopen(argv[11], "wb");
  if (!fo) { perror(argv[11]); return 2; }
  jpeg_stdio_dest(&c, fo);
  c.image_width = W; c.image_height = H; c.input_components = 3;
  c.in_color_space = ycc ? JCS_YCbCr : JCS_RGB;
  jpeg_set_defaults(&c);
  jpeg_set_quality(&c, quality, TRUE);
  c.comp_info[0].h_samp_factor = hs; c.comp_info[0].v_samp_factor = vs;
  c.comp_info[1].h_samp_factor = 1; c.comp_info[1].v_samp_factor = 1;
  c.comp_info[2].h_samp_factor = 1; c.comp_info[2].v_samp_factor = 1;
  c.optimize_coding = opt ? TRUE : FALSE;
  c.restart_interval = rst;
  c.dct_method = JDCT_ISLOW;
#if JPEG_LIB_VERSION >= 70
  c.do_fancy_downsampling = FALSE; /* spatial box downsampling, as in T.81-era encoders (and libjpeg-turbo) */
#endif
  jpeg_start_compress(&c, TRUE);
  while (c.next_scanline < c.image_height) {
    JSAMPROW row = raw + (size_t)c.next_scanline * W * 3;
    jpeg_write_scanlines(&c, &row, 1);
  }
  jpeg_finish_compress(&c);
  jpeg_destroy_compress(&c);
  fclose(fo); free(raw);
  return 0;
}

#include <time.h>
static int do_bench(int argc, char **argv) {
  if (argc != 11) return 1;
  size_t n; unsigned char *raw = slurp(argv[2], &n);
  int W = atoi(argv[3]), H = atoi(argv[4]);
  int quality = atoi(argv[5]), hs = atoi(argv[6]), vs = atoi(argv[7]), opt = atoi(argv[8]), rst = atoi(argv[9]), reps = atoi(argv[10]);
  if (n < (size_t)W * H * 3) { fprintf(stderr, "short input\n"); return 2; }
  double best = 1e30; unsigned long bytes = 0;
  for (int r = 0; r < reps; r++) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    struct jpeg_compress_struct c; struct jpeg_error_mgr e;
    c.err = jpeg_std_error(&e);
    jpeg_create_compress(&c);
    unsigned char *out = NULL; unsigned long outn = 0;
    jpeg_mem_dest(&c, &out, &outn);
    c.image_width = W; c.image_height = H; c.input_components = 3; c.in_color_space = JCS_RGB;
    jpeg_set_defaults(&c);
    jpeg_set_quality(&c, quality, TRUE);
    c.comp_info[0].h_samp_factor = hs; c.comp_info[0].v_samp_factor = vs;
    c.comp_info[1].h_samp_factor = 1; c.comp_info[1].v_samp_factor = 1;
    c.comp_info[2].h_samp_factor = 1; c.comp_info[2].v_samp_factor = 1;
    c.optimize_coding = opt ? TRUE : FALSE;
    c.restart_interval = rst;
    c.dct_method = JDCT_ISLOW;
#if JPEG_LIB_VERSION >= 70
    c.do_fancy_downsampling = FALSE;
#endif
    jpeg_start_compress(&c, TRUE);
    while (c.next_scanline < c.image_height) {
      JSAMPROW row = raw + (size_t)c.next_scanline * W * 3;
      jpeg_write_scanlines(&c, &row, 1);
    }
    jpeg_finish_compress(&c);
    jpeg_destroy_compress(&c);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    if (dt < best) best = dt;
    bytes = outn;
    free(out);
  }
  printf("%.6f %lu\n", best, bytes);
  free(raw);
  return 0;
}

static int do_coef(int argc, char **argv) {
  if (argc != 4) return 1;
  size_t n; unsigned char *jpg = slurp(argv[2], &n);
  struct jpeg_decompress_struct d; struct jpeg_error_mgr e;
  d.err = jpeg_std_error(&e);
  jpeg_create_decompress(&d);
  jpeg_mem_src(&d, jpg, (unsigned long)n);
  jpeg_read_header(&d, TRUE);
  jvirt_barray_ptr *arr = jpeg_read_coefficients(&d);
  FILE *fo = fopen(argv[3], "wb");
  int32_t hdr = d.num_components; fwrite(&hdr, 4, 1, fo);
  for (int ci = 0; ci < d.num_components; ci++) {
    jpeg_component_info *ci_ = &d.comp_info[ci];
    int32_t h4[4] = {(int32_t)ci_->width_in_blocks, (int32_t)ci_->height_in_blocks, ci_->h_samp_factor, ci_->v_samp_factor};
    fwrite(h4, 4, 4, fo);
  }
  for (int ci = 0; ci < d.num_components; ci++) {
    jpeg_component_info *ci_ = &d.comp_info[ci];
    for (JDIMENSION by = 0; by < ci_->height_in_blocks; by++) {
      JBLOCKARRAY rows = (*d.mem->access_virt_barray)((j_common_ptr)&d, arr[ci], by, 1, FALSE);
      for (JDIMENSION bx = 0; bx < ci_->width_in_blocks; bx++) {
        int16_t tmp[64];
        for (int k = 0; k < 64; k++) tmp[k] = (int16_t)rows[0][bx][k];
        fwrite(tmp, 2, 64, fo);
      }
    }
  }
  fclose(fo);
  jpeg_finish_decompress(&d);
  jpeg_destroy_decompress(&d);
  free(jpg);
  return 0;
}

static int do_dec(int argc, char **argv) {
  if (argc != 4) return 1;
  size_t n; unsigned char *jpg = slurp(argv[2], &n);
  struct jpeg_decompress_struct d; struct jpeg_error_mgr e;
  d.err = jpeg_std_error(&e);
  jpeg_create_decompress(&d);
  jpeg_mem_src(&d, jpg, (unsigned long)n);
  jpeg_read_header(&d, TRUE);
  d.out_color_space = JCS_RGB;
  jpeg_start_decompress(&d);
  FILE *fo = fopen(argv[3], "wb");
  unsigned char *row = (unsigned char *)malloc((size_t)d.output_width * 3);
  while (d.output_scanline < d.output_height) {
    JSAMPROW r = row;
    jpeg_read_scanlines(&d, &r, 1);
    fwrite(row, 1, (size_t)d.output_width * 3, fo);
  }
  fclose(fo); free(row);
  jpeg_finish_decompress(&d);
  jpeg_destroy_decompress(&d);
  free(jpg);
  return 0;
}

int main(int argc, char **argv) {
  int rc = 1;
  if (argc >= 2) {
    if (!strcmp(argv[1], "enc")) rc = do_enc(argc, argv);
    else if (!strcmp(argv[1], "coef")) rc = do_coef(argc, argv);
    else if (!strcmp(argv[1], "dec")) rc = do_dec(argc, argv);
    else if (!strcmp(argv[1], "bench")) rc = do_bench(argc, argv);
  }
  if (rc == 1) fprintf(stderr, "usage: see header comment of ijg_harness.c\n");
  return rc;
}

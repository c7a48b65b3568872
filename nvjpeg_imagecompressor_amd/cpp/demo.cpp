// demo.cpp -- the reference demo's sequence (reference src/ImageCompressor/main.cpp:16-81) on portable inputs:
//   new NvjpegCompressRunner -> buildCompressEnv -> compress x2 -> deleteCompressEnv -> buildDecodeEnv -> save x2 ->
//   decode x2 -> deleteDecodeEnv -> delete.
// Inputs are binary PPM (P6) files, or a synthetic image when no path is given (no OpenCV imread in this image).
//   demo [in1.ppm [in2.ppm]] [--css N] [--quality Q] [--out prefix]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/ImageCompressor.h"

static void printState(int run_state) { std::cout << (run_state == 1 ? "[INFO] Successful." : "[INFO] Failed.") << std::endl; }

static cv::Mat readPPM(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) return cv::Mat();
  int w = 0, h = 0, maxv = 0;
  char magic[3] = {0};
  if (fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) != 4 || strcmp(magic, "P6") || maxv != 255) { fclose(f); return cv::Mat(); }
  fgetc(f);
  cv::Mat m(h, w, CV_8UC3);
  for (int y = 0; y < h; y++) {
    unsigned char *row = m.ptr<unsigned char>(y);
    if (fread(row, 3, (size_t)w, f) != (size_t)w) { fclose(f); return cv::Mat(); }
    for (int x = 0; x < w; x++) { unsigned char t = row[3 * x]; row[3 * x] = row[3 * x + 2]; row[3 * x + 2] = t; }  // RGB -> BGR
  }
  fclose(f);
  return m;
}

static cv::Mat synthetic(int w, int h, int phase) {
  cv::Mat m(h, w, CV_8UC3);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      unsigned char *p = m.ptr<unsigned char>(y) + 3 * x;
      p[0] = (unsigned char)((x + phase) & 255); p[1] = (unsigned char)((y * 2 + phase) & 255); p[2] = (unsigned char)(((x ^ y) + phase) & 255);
    }
  return m;
}

int main(int argc, char *argv[]) {
  int css = 0, quality = 95;
  std::string out = "demo_out";
  const char *in[2] = {nullptr, nullptr};
  int nin = 0;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--css") && i + 1 < argc) css = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--quality") && i + 1 < argc) quality = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--out") && i + 1 < argc) out = argv[++i];
    else if (nin < 2) in[nin++] = argv[i];
  }
  cv::Mat image1 = in[0] ? readPPM(in[0]) : synthetic(1040, 520, 0);
  cv::Mat image2 = in[1] ? readPPM(in[1]) : synthetic(image1.cols, image1.rows, 77);
  if (image1.empty() || image2.empty() || image2.cols != image1.cols || image2.rows != image1.rows) {
    std::cerr << "could not read the inputs (two P6 PPMs of equal size)" << std::endl;
    return EXIT_FAILURE;
  }
  NvjpegCompressRunner *compressor = new NvjpegCompressRunner(image1.cols, image1.rows, quality, true);
  compressor->setSamplingFactors(css);
  int compress_run_state = 0;
  compressor->buildCompressEnv();
  std::vector<unsigned char> obuffer1 = compressor->compress(image1, &compress_run_state);
  printState(compress_run_state);
  int ok = compress_run_state;
  std::vector<unsigned char> obuffer2 = compressor->compress(image2, &compress_run_state);
  printState(compress_run_state);
  ok &= compress_run_state;
  compressor->deleteCompressEnv();
  compressor->buildDecodeEnv();
  compressor->save(out + "_1.jpeg", obuffer1);
  compressor->save(out + "_2.jpeg", obuffer2);
  // decode both files back (reference main.cpp:65-75) and write them as PPM (no cv::imwrite here)
  for (int i = 1; i <= 2 && ok; i++) {
    int decode_run_state = 0;
    cv::Mat dec = compressor->decode(out + "_" + std::to_string(i) + ".jpeg", &decode_run_state);
    printState(decode_run_state);
    ok &= decode_run_state;
    if (decode_run_state) {
      FILE *f = fopen((out + "_" + std::to_string(i) + "_decode.ppm").c_str(), "wb");
      if (f) {
        fprintf(f, "P6\n%d %d\n255\n", dec.cols, dec.rows);
        for (int y = 0; y < dec.rows; y++)
          for (int x = 0; x < dec.cols; x++) { const unsigned char *p = dec.ptr<unsigned char>(y) + 3 * x; unsigned char rgb[3] = {p[2], p[1], p[0]}; fwrite(rgb, 1, 3, f); }
        fclose(f);
      }
    }
  }
  compressor->deleteDecodeEnv();
  delete compressor;
  return ok ? EXIT_SUCCESS : EXIT_FAILURE;
}

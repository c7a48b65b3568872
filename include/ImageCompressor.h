// ImageCompressor.h -- source-compatible drop-in for the reference's public facade
// (reference src/ImageCompressorDll/ImageCompressor.h:22-42): same class name, constructor defaults and methods.
// Underneath, every call goes to the C ABI in include/mi_jpeg.h (HIP kernels for MI355X); there is no nvJPEG, no CUDA.
//
// Behavioural contract kept from the reference (ImageCompressor.cpp:45-101):
//   * compress() returns a complete JFIF file; an EMPTY vector means failure; *run_state = 1 / 0; run_state may be null.
//   * decode() takes a file path and returns a BGR CV_8UC3 Mat of the file's own size; empty Mat on failure.
//   * the stdout lines "=> Compress Cost time : X ms" and "[INFO] NvjpegCompressRunner Compress Func Cost Time : N ms".
// Deliberate differences: nothing ever calls exit() (the reference's CHECK_CUDA / CHECK_NVJPEG do,
// ImageCompressorImpl.cuh:16-34); an image whose size differs from the constructor's is refused instead of overrunning
// the device planes (reference ImageCompressorImpl.cu:275,280); buildCompressEnv() twice is a no-op instead of a leak;
// the output is baseline sequential (SOF0) with restart intervals instead of progressive (reference .cu:28).
// Additive extensions: chroma subsampling, restart interval and optimised-Huffman setters (take effect at the next
// buildCompressEnv), verbosity switch.
#ifndef IMAGECOMPRESSOR_H_
#define IMAGECOMPRESSOR_H_

#include <iostream>
#include <string>
#include <vector>

#include "cvmat_min.h"

#define NVJPEG_COMPRESS_RUNNER_API __attribute__((visibility("default")))

class NvjpegCompressRunnerImpl;

class NVJPEG_COMPRESS_RUNNER_API NvjpegCompressRunner {
 private:
  NvjpegCompressRunnerImpl *compressor;

 public:
  NvjpegCompressRunner(int width = 8320, int height = 40000, int quality = 95, bool optimize = true);
  ~NvjpegCompressRunner();

  NvjpegCompressRunner(const NvjpegCompressRunner &) = delete;
  NvjpegCompressRunner &operator=(const NvjpegCompressRunner &) = delete;

  std::vector<unsigned char> compress(cv::Mat image, int *run_state);
  cv::Mat decode(std::string image_path, int *run_state);
  void save(std::string save_path, std::vector<unsigned char> obuffer);

  void buildCompressEnv();
  void buildDecodeEnv();
  void deleteCompressEnv();
  void deleteDecodeEnv();

  // ---- extensions (defaults reproduce the reference: 4:4:4, library-chosen restart interval) ----
  void setSamplingFactors(int css);        // nvjpegChromaSubsampling_t values: 0=444 1=422 2=420 3=440 4=411 5=410
  void setQuality(int quality);
  void setOptimizedHuffman(bool optimize);
  void setRestartInterval(int mcus);       // -1 = automatic
  void setProgressive(bool progressive);   // the reference's encoding (ImageCompressorImpl.cu:28); default false = baseline, the fast path
  // Secondary ("difference map") compression, reference README.md:8: first layer into `primary`, the
  // JPEG of the difference map is returned. Needs the compress environment only (the first layer's reconstruction comes from
  // the encoder's own coefficients). secondaryDecode puts the pair back together (decode environment).
  // The overloads give the second layer its own quality (1..100; 0 = the first layer's), sampling (css as above; -1 = the first
  // layer's) and a gain (1, 2, 4, 8) applied to the difference before it is coded -- mij_secondary_params in mi_jpeg.h; the
  // same gain must be passed to secondaryDecode.
  std::vector<unsigned char> secondaryCompress(cv::Mat image, std::vector<unsigned char> &primary, int *run_state);
  std::vector<unsigned char> secondaryCompress(cv::Mat image, std::vector<unsigned char> &primary, int quality2, int css2, int gain, int *run_state);
  cv::Mat secondaryDecode(const std::vector<unsigned char> &primary, const std::vector<unsigned char> &secondary, int *run_state);
  cv::Mat secondaryDecode(const std::vector<unsigned char> &primary, const std::vector<unsigned char> &secondary, int gain, int *run_state);
  void setDevice(int device);
  void setVerbose(bool verbose);
  const char *lastError() const;
};

using ImageCompressor = NvjpegCompressRunner;   // neutral alias

#endif  // IMAGECOMPRESSOR_H_

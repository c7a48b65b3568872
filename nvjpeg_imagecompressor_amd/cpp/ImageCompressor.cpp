// ImageCompressor.cpp -- facade implementation over the mi_jpeg C ABI (see ImageCompressor.h for the contract).
#include "ImageCompressor.h"

#include <chrono>
#include <cstdio>
#include <fstream>

#include "../../include/mi_jpeg.h"

class NvjpegCompressRunnerImpl {
 public:
  mij_encoder_params p{};
  mij_encoder *enc = nullptr;
  bool verbose = true, decode_init = false;
  std::string err;
};

NvjpegCompressRunner::NvjpegCompressRunner(int width, int height, int quality, bool optimize) {
  compressor = new NvjpegCompressRunnerImpl();
  compressor->p.width = width; compressor->p.height = height; compressor->p.quality = quality;
  compressor->p.optimized_huffman = optimize ? 1 : 0;
  compressor->p.css = MIJ_CSS_444;                 // the reference hard-codes 4:4:4 (ImageCompressorImpl.cu:31)
  compressor->p.restart_interval = MIJ_RESTART_AUTO;
}

NvjpegCompressRunner::~NvjpegCompressRunner() {
  deleteCompressEnv();
  const bool v = compressor->verbose;
  delete compressor;
  if (v) std::cout << "[INFO] Delete NvjpegCompressRunnerImpl Successfully ..." << std::endl;
}

void NvjpegCompressRunner::buildCompressEnv() {
  if (compressor->enc) return;
  if (mij_encoder_create(&compressor->p, &compressor->enc) != MIJ_OK) {
    compressor->err = mij_last_error(nullptr);
    compressor->enc = nullptr;
    return;
  }
  mij_encoder_enable_timing(compressor->enc, 1);
}
void NvjpegCompressRunner::deleteCompressEnv() { mij_encoder_destroy(compressor->enc); compressor->enc = nullptr; }
void NvjpegCompressRunner::buildDecodeEnv() { compressor->decode_init = true; }
void NvjpegCompressRunner::deleteDecodeEnv() { compressor->decode_init = false; }

std::vector<unsigned char> NvjpegCompressRunner::compress(cv::Mat image, int *run_state) {
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<unsigned char> obuffer;
  mij_encoder *e = compressor->enc;
  if (!e) {
    std::cerr << "[ERROR] compress() called before buildCompressEnv() succeeded: " << compressor->err << std::endl;
  } else if (image.empty() || image.type() != CV_8UC3 || image.cols != compressor->p.width || image.rows != compressor->p.height) {
    std::cerr << "[ERROR] compress(): image must be CV_8UC3 " << compressor->p.width << "x" << compressor->p.height << std::endl;
  } else {
    const uint8_t *jpg = nullptr;
    size_t n = 0;
    if (mij_encode_host(e, image.ptr<unsigned char>(0), image.step, 0, MIJ_INPUT_BGRI, &jpg, &n) == MIJ_OK) {
      obuffer.assign(jpg, jpg + n);
      float ms[MIJ_NUM_STAGE_TIMES];
      if (compressor->verbose && mij_stage_times(e, ms) == MIJ_OK) std::cout << "=> Compress Cost time : " << ms[6] << "ms" << std::endl;
    } else {
      compressor->err = mij_last_error(e);
    }
  }
  if (run_state) *run_state = obuffer.empty() ? 0 : 1;
  const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  if (compressor->verbose) std::cout << "[INFO] NvjpegCompressRunner Compress Func Cost Time : " << ms << " ms" << std::endl;
  return obuffer;
}

cv::Mat NvjpegCompressRunner::decode(std::string image_path, int *run_state) {
  // The decode path (reference ImageCompressorImpl.cu:311-385) is the next row of the scope table; until its HIP
  // kernels land this reports failure the reference's way (empty Mat, run_state 0) rather than decoding on the CPU.
  FILE *f = fopen(image_path.c_str(), "rb");
  if (!f) {
    std::cerr << "Failed to open JPEG file." << std::endl;
  } else {
    fclose(f);
    std::cerr << "[ERROR] decode(): HIP decoder not built in this version" << std::endl;
  }
  if (run_state) *run_state = 0;
  return cv::Mat();
}

void NvjpegCompressRunner::save(std::string save_path, std::vector<unsigned char> obuffer) {
  try {
    std::ofstream outputFile(save_path, std::ios::out | std::ios::binary);
    outputFile.write(reinterpret_cast<const char *>(obuffer.data()), static_cast<std::streamsize>(obuffer.size()));
    outputFile.close();
  } catch (const std::exception &e) {
    std::cerr << "Exception caught: " << e.what() << std::endl;
  }
}

void NvjpegCompressRunner::setSamplingFactors(int css) { compressor->p.css = css; }
void NvjpegCompressRunner::setQuality(int q) { compressor->p.quality = q; }
void NvjpegCompressRunner::setOptimizedHuffman(bool o) { compressor->p.optimized_huffman = o ? 1 : 0; }
void NvjpegCompressRunner::setRestartInterval(int m) { compressor->p.restart_interval = m; }
void NvjpegCompressRunner::setDevice(int d) { compressor->p.device = d; }
void NvjpegCompressRunner::setVerbose(bool v) { compressor->verbose = v; }
const char *NvjpegCompressRunner::lastError() const { return compressor->err.c_str(); }

// ImageCompressor.cpp -- facade implementation over the mi_jpeg C ABI (see ImageCompressor.h for the contract).
#include "../../include/ImageCompressor.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fstream>

#include "../../include/mi_jpeg.h"

class NvjpegCompressRunnerImpl {
 public:
  mij_encoder_params p = MIJ_ENCODER_PARAMS_INIT;
  mij_encoder *enc = nullptr;
  mij_decoder *dec = nullptr;
  bool verbose = true;
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
  deleteDecodeEnv();
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
void NvjpegCompressRunner::buildDecodeEnv() {
  if (compressor->dec) return;
  if (mij_decoder_create(compressor->p.device, &compressor->dec) != MIJ_OK) { compressor->err = mij_decoder_last_error(nullptr); compressor->dec = nullptr; }
}
void NvjpegCompressRunner::deleteDecodeEnv() { mij_decoder_destroy(compressor->dec); compressor->dec = nullptr; }

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
  // reference ImageCompressor.cpp:63-88 + ImageCompressorImpl.cu:311-385: open, read whole file, decode, BGR Mat.
  cv::Mat result;
  FILE *jpeg_file = fopen(image_path.c_str(), "rb");
  if (!jpeg_file) {
    std::cerr << "Failed to open JPEG file." << std::endl;
    if (run_state) *run_state = 0;
    return cv::Mat();
  }
  const auto t0 = std::chrono::steady_clock::now();
  fseek(jpeg_file, 0, SEEK_END);
  const long sz = ftell(jpeg_file);
  rewind(jpeg_file);
  std::vector<unsigned char> data(sz > 0 ? (size_t)sz : 0);
  const size_t got = data.empty() ? 0 : fread(data.data(), 1, data.size(), jpeg_file);
  fclose(jpeg_file);
  if (data.empty() || got != data.size()) {
    std::cerr << "[INFO] Failed to read the entire JPEG data." << std::endl;
  } else if (!compressor->dec) {
    std::cerr << "[ERROR] decode() called before buildDecodeEnv() succeeded: " << compressor->err << std::endl;
  } else {
    int w = 0, h = 0;
    if (mij_decode_info(data.data(), data.size(), &w, &h, nullptr, nullptr) == MIJ_OK) {
      cv::Mat m(h, w, CV_8UC3);
      if (mij_decode_host(compressor->dec, data.data(), data.size(), m.ptr<unsigned char>(0), m.step, MIJ_INPUT_BGRI, &w, &h) == MIJ_OK) {
        result = m;
        float dms = 0;   // reference ImageCompressorImpl.cu:373 (whose start event is never recorded; this one is measured)
        if (compressor->verbose && mij_decode_last_ms(compressor->dec, &dms) == MIJ_OK) std::cout << "=> Decode Cost time : " << dms << "ms" << std::endl;
      } else
        compressor->err = mij_decoder_last_error(compressor->dec);
    } else {
      compressor->err = mij_decoder_last_error(nullptr);
      std::cerr << "[ERROR] JPEG decode failed: " << compressor->err << std::endl;
    }
  }
  if (run_state) *run_state = result.empty() ? 0 : 1;
  const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  if (compressor->verbose) std::cout << "[INFO] NvjpegCompressRunner Decode Func Cost Time : " << ms << " ms" << std::endl;
  return result;
}

std::vector<unsigned char> NvjpegCompressRunner::secondaryCompress(cv::Mat image, std::vector<unsigned char> &primary, int *run_state) {
  return secondaryCompress(image, primary, 0, -1, 1, run_state);
}

std::vector<unsigned char> NvjpegCompressRunner::secondaryCompress(cv::Mat image, std::vector<unsigned char> &primary, int quality2, int css2, int gain,
                                                                   int *run_state) {
  std::vector<unsigned char> secondary;
  primary.clear();
  mij_secondary_params sp = MIJ_SECONDARY_PARAMS_INIT;
  sp.quality2 = quality2; sp.css2 = css2; sp.gain = gain;
  if (!compressor->enc) {
    std::cerr << "[ERROR] secondaryCompress() called before buildCompressEnv() succeeded" << std::endl;
  } else if (image.empty() || image.type() != CV_8UC3 || image.cols != compressor->p.width || image.rows != compressor->p.height) {
    std::cerr << "[ERROR] secondaryCompress(): image must be CV_8UC3 " << compressor->p.width << "x" << compressor->p.height << std::endl;
  } else {
    // Start from a generous guess and let the library say what it needs: a JPEG is usually far below the raw size, but a
    // residual image is close to noise and, at q100 with dense restart markers, a layer can exceed it (MIJ_ERR_OVERFLOW
    // then reports the required sizes).
    size_t cap1 = (size_t)image.cols * image.rows * 3 / 2 + 65536, cap2 = cap1;
    int rc = MIJ_ERR_OVERFLOW;
    for (int attempt = 0; attempt < 4 && rc == MIJ_ERR_OVERFLOW; attempt++) {
      primary.resize(cap1); secondary.resize(cap2);
      size_t n1 = cap1, n2 = cap2;
      rc = mij_secondary_encode_host_ex(compressor->enc, &sp, image.ptr<unsigned char>(0), image.step, 0, MIJ_INPUT_BGRI, primary.data(), &n1,
                                        secondary.data(), &n2);
      if (rc == MIJ_OK) { primary.resize(n1); secondary.resize(n2); }
      else if (rc == MIJ_ERR_OVERFLOW) { cap1 = std::max(cap1, n1 + n1 / 8 + 4096); cap2 = std::max(cap2, n2 + n2 / 8 + 4096); if (n2 == 0) cap2 = std::max(cap2, cap1); }
    }
    if (rc != MIJ_OK) {
      compressor->err = mij_last_error(compressor->enc);
      primary.clear(); secondary.clear();
    }
  }
  if (run_state) *run_state = secondary.empty() ? 0 : 1;
  return secondary;
}

cv::Mat NvjpegCompressRunner::secondaryDecode(const std::vector<unsigned char> &primary, const std::vector<unsigned char> &secondary, int *run_state) {
  return secondaryDecode(primary, secondary, 1, run_state);
}

cv::Mat NvjpegCompressRunner::secondaryDecode(const std::vector<unsigned char> &primary, const std::vector<unsigned char> &secondary, int gain,
                                              int *run_state) {
  cv::Mat result;
  int w = 0, h = 0;
  mij_secondary_params sp = MIJ_SECONDARY_PARAMS_INIT;
  sp.gain = gain;
  if (!compressor->dec) {
    std::cerr << "[ERROR] secondaryDecode() called before buildDecodeEnv() succeeded" << std::endl;
  } else if (mij_decode_info(primary.data(), primary.size(), &w, &h, nullptr, nullptr) == MIJ_OK) {
    cv::Mat m(h, w, CV_8UC3);
    if (mij_secondary_decode_host_ex(compressor->dec, &sp, primary.data(), primary.size(), secondary.data(), secondary.size(), m.ptr<unsigned char>(0),
                                     m.step, MIJ_INPUT_BGRI, &w, &h) == MIJ_OK)
      result = m;
    else
      compressor->err = mij_decoder_last_error(compressor->dec);
  }
  if (run_state) *run_state = result.empty() ? 0 : 1;
  return result;
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
void NvjpegCompressRunner::setProgressive(bool on) { compressor->p.progressive = on ? 1 : 0; }
void NvjpegCompressRunner::setDevice(int d) { compressor->p.device = d; }
void NvjpegCompressRunner::setVerbose(bool v) { compressor->verbose = v; }
const char *NvjpegCompressRunner::lastError() const { return compressor->err.c_str(); }

// mij_decode_api.hip -- host side of the decode half of the C ABI (include/mi_jpeg.h): marker parsing and table
// construction on the host (what nvjpegJpegStreamParse / nvjpegGetImageInfo do, reference ImageCompressorImpl.cu:335,362),
// everything else on the device (k_decode.inc). No CPU decode fallback.
#include "../../include/mi_jpeg.h"
#include "mij_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

using namespace mij;

struct mij_decoder {
  int device = 0;
  std::string err;
  uint8_t *d_scan = nullptr; size_t scan_cap = 0;
  int16_t *d_coef = nullptr; size_t coef_cap = 0;
  uint8_t *d_planes = nullptr; size_t planes_cap = 0;
  DecTables *d_tab = nullptr;
  unsigned long long *d_seg_pos = nullptr; size_t seg_cap = 0;
  unsigned long long *d_chunk_cnt = nullptr, *d_chunk_base = nullptr; size_t chunk_cap = 0;
  uint32_t *d_flags = nullptr;    // [0] scratch for the scan kernel, [1] Huffman decode errors
  DeviceResult *d_res = nullptr;
  uint8_t *d_out = nullptr; size_t out_cap = 0;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev0{}, ev1{};
  bool ev_ok = false, issued = false;
};

static thread_local std::string g_dec_err;
static int dfail(mij_decoder *d, int code, const char *what, hipError_t he = hipSuccess) {
  std::string m = what;
  if (he != hipSuccess) { m += ": "; m += hipGetErrorString(he); }
  if (d) d->err = m; else g_dec_err = m;
  fprintf(stderr, "[ERROR] mi_jpeg: %s\n", m.c_str());
  return code;
}
#define DHIP(d, x) do { hipError_t he_ = (x); if (he_ != hipSuccess) return dfail((d), MIJ_ERR_HIP, #x, he_); } while (0)

static const uint8_t kZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Parsed {
  int W = 0, H = 0, hs = 0, vs = 0, ri = 0;
  DecTables t{};
  bool have_tab[4] = {false, false, false, false};
  size_t scan_off = 0;
};

// Baseline sequential, 3 components, chroma 1x1, luma h x v with h in {1,2,4}, v in {1,2}: everything this project's
// encoder writes (and what libjpeg writes for the common samplings). Anything else -> MIJ_ERR_BAD_STREAM.
static int parse_jpeg(const uint8_t *p, size_t n, Parsed &o, std::string &why) {
  if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) { why = "not a JPEG (no SOI)"; return MIJ_ERR_BAD_STREAM; }
  size_t i = 2;
  while (i + 4 <= n) {
    if (p[i] != 0xFF) { why = "marker expected"; return MIJ_ERR_BAD_STREAM; }
    const int m = p[i + 1];
    if (m == 0xFF) { i++; continue; }
    const size_t len = ((size_t)p[i + 2] << 8) | p[i + 3];
    if (len < 2 || i + 2 + len > n) { why = "truncated segment"; return MIJ_ERR_BAD_STREAM; }
    const uint8_t *s = p + i + 4;
    const size_t pl = len - 2;
    if (m == 0xDB) {
      for (size_t k = 0; k + 65 <= pl; k += 65) {
        if ((s[k] >> 4) != 0 || (s[k] & 15) > 1) { why = "unsupported DQT"; return MIJ_ERR_BAD_STREAM; }
        for (int z = 0; z < 64; z++) o.t.q[s[k] & 15][kZZ[z]] = s[k + 1 + z];
      }
    } else if (m == 0xC0) {
      if (pl < 15 || s[0] != 8 || s[5] != 3) { why = "only 8-bit 3-component frames"; return MIJ_ERR_BAD_STREAM; }
      o.H = (s[1] << 8) | s[2]; o.W = (s[3] << 8) | s[4];
      for (int c = 0; c < 3; c++) {
        const int hv = s[7 + 3 * c];
        o.t.tq[c] = s[8 + 3 * c] & 1;
        if (c == 0) { o.hs = hv >> 4; o.vs = hv & 15; } else if (hv != 0x11) { why = "chroma sampling must be 1x1"; return MIJ_ERR_BAD_STREAM; }
      }
    } else if (m >= 0xC1 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      why = "only baseline sequential (SOF0) is handled";
      return MIJ_ERR_BAD_STREAM;
    } else if (m == 0xC4) {
      size_t k = 0;
      while (k + 17 <= pl) {
        const int tc = s[k] >> 4, th = s[k] & 15;
        if (tc > 1 || th > 1) { why = "unsupported DHT"; return MIJ_ERR_BAD_STREAM; }
        const int t = th * 2 + tc;
        int cnt = 0;
        for (int l = 1; l <= 16; l++) cnt += s[k + l];
        if (cnt > 256 || k + 17 + (size_t)cnt > pl) { why = "bad DHT"; return MIJ_ERR_BAD_STREAM; }
        // canonical codes -> look-ahead + slow-path tables
        memset(o.t.look[t], 0, sizeof o.t.look[t]);
        memcpy(o.t.vals[t], s + k + 17, (size_t)cnt);
        int code = 0, idx = 0;
        for (int l = 1; l <= 16; l++) {
          const int nb = s[k + l];
          o.t.valoff[t][l] = idx - code;
          for (int j = 0; j < nb; j++, code++, idx++) {
            if (l <= 9) {
              const int lo = code << (9 - l);
              for (int f = 0; f < (1 << (9 - l)); f++) o.t.look[t][lo + f] = (uint16_t)((l << 8) | s[k + 17 + idx]);
            }
          }
          o.t.maxcode[t][l] = nb ? code - 1 : -1;
          code <<= 1;
        }
        o.t.maxcode[t][0] = -1; o.t.maxcode[t][17] = 0x7FFFFFFF; o.t.valoff[t][0] = 0;
        o.have_tab[t] = true;
        k += 17 + (size_t)cnt;
      }
    } else if (m == 0xDD) {
      if (pl >= 2) o.ri = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (pl < 10 || s[0] != 3) { why = "only interleaved 3-component scans"; return MIJ_ERR_BAD_STREAM; }
      for (int c = 0; c < 3; c++) { o.t.td[c] = (s[2 + 2 * c] >> 4) & 1; o.t.ta[c] = (s[2 + 2 * c] & 15) & 1; }
      o.scan_off = i + 2 + len;
      if (o.W <= 0 || o.H <= 0) { why = "SOS before SOF"; return MIJ_ERR_BAD_STREAM; }
      if (!((o.hs == 1 || o.hs == 2 || o.hs == 4) && (o.vs == 1 || o.vs == 2))) { why = "unsupported luma sampling"; return MIJ_ERR_BAD_STREAM; }
      for (int c = 0; c < 3; c++)
        if (!o.have_tab[o.t.td[c] * 2] || !o.have_tab[o.t.ta[c] * 2 + 1]) { why = "missing Huffman table"; return MIJ_ERR_BAD_STREAM; }
      return MIJ_OK;
    }
    i += 2 + len;
  }
  why = "no SOS";
  return MIJ_ERR_BAD_STREAM;
}

static int css_of(int hs, int vs) {
  if (hs == 1 && vs == 1) return MIJ_CSS_444;
  if (hs == 2 && vs == 1) return MIJ_CSS_422;
  if (hs == 2 && vs == 2) return MIJ_CSS_420;
  if (hs == 1 && vs == 2) return MIJ_CSS_440;
  if (hs == 4 && vs == 1) return MIJ_CSS_411;
  if (hs == 4 && vs == 2) return MIJ_CSS_410;
  return -1;
}

template <class T>
static int ensure(mij_decoder *d, T *&ptr, size_t &cap, size_t need) {
  if (need <= cap) return MIJ_OK;
  (void)hipFree(ptr); ptr = nullptr; cap = 0;
  hipError_t he = hipMalloc(&ptr, need * sizeof(T));
  if (he != hipSuccess) return dfail(d, MIJ_ERR_ALLOC, "hipMalloc (decoder workspace)", he);
  cap = need;
  return MIJ_OK;
}

extern "C" {

int mij_decode_info(const uint8_t *jpeg, size_t jpeg_bytes, int *width, int *height, int *css, int *restart_interval) {
  if (!jpeg) return MIJ_ERR_INVALID_ARG;
  Parsed ps; std::string why;
  int rc = parse_jpeg(jpeg, jpeg_bytes, ps, why);
  if (rc) { g_dec_err = why; return rc; }
  if (width) *width = ps.W;
  if (height) *height = ps.H;
  if (css) *css = css_of(ps.hs, ps.vs);
  if (restart_interval) *restart_interval = ps.ri;
  return MIJ_OK;
}

const char *mij_decoder_last_error(const mij_decoder *dec) { return dec ? dec->err.c_str() : g_dec_err.c_str(); }

void mij_decoder_destroy(mij_decoder *d) {
  if (!d) return;
  (void)hipSetDevice(d->device);
  if (d->issued) (void)hipStreamSynchronize(d->last_stream);
  (void)hipFree(d->d_scan); (void)hipFree(d->d_coef); (void)hipFree(d->d_planes); (void)hipFree(d->d_tab);
  (void)hipFree(d->d_seg_pos); (void)hipFree(d->d_chunk_cnt); (void)hipFree(d->d_chunk_base); (void)hipFree(d->d_flags);
  (void)hipFree(d->d_res); (void)hipFree(d->d_out);
  if (d->ev_ok) { (void)hipEventDestroy(d->ev0); (void)hipEventDestroy(d->ev1); }
  delete d;
}

int mij_decoder_create(int device, mij_decoder **out) {
  if (!out) return MIJ_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return dfail(nullptr, MIJ_ERR_NO_DEVICE, "no HIP device: mi_jpeg has no CPU fallback");
  if (device < 0 || device >= ndev) return dfail(nullptr, MIJ_ERR_INVALID_ARG, "device ordinal out of range");
  DHIP(nullptr, hipSetDevice(device));
  mij_decoder *d = new (std::nothrow) mij_decoder();
  if (!d) return dfail(nullptr, MIJ_ERR_ALLOC, "out of host memory");
  d->device = device;
  hipError_t he;
  if ((he = hipMalloc(&d->d_tab, sizeof(DecTables))) != hipSuccess || (he = hipMalloc(&d->d_flags, 2 * sizeof(uint32_t))) != hipSuccess ||
      (he = hipMalloc(&d->d_res, sizeof(DeviceResult))) != hipSuccess || (he = hipMemset(d->d_flags, 0, 2 * sizeof(uint32_t))) != hipSuccess ||
      (he = hipEventCreate(&d->ev0)) != hipSuccess || (he = hipEventCreate(&d->ev1)) != hipSuccess) {
    mij_decoder_destroy(d);
    return dfail(nullptr, MIJ_ERR_HIP, "decoder setup", he);
  }
  d->ev_ok = true;
  *out = d;
  return MIJ_OK;
}

int mij_decode_device(mij_decoder *d, const uint8_t *jpeg, size_t jpeg_bytes, void *d_dst, size_t pitch, size_t plane_stride,
                      int output_format, void *stream) {
  if (!d || !jpeg || !d_dst) return dfail(d, MIJ_ERR_INVALID_ARG, "null argument");
  const bool interleaved = output_format == MIJ_INPUT_BGRI || output_format == MIJ_INPUT_RGBI;
  if (!interleaved && output_format != MIJ_INPUT_BGR && output_format != MIJ_INPUT_RGB) return dfail(d, MIJ_ERR_INVALID_ARG, "unknown output format");
  Parsed ps; std::string why;
  int rc = parse_jpeg(jpeg, jpeg_bytes, ps, why);
  if (rc) return dfail(d, rc, why.c_str());
  if (pitch < (size_t)ps.W * (interleaved ? 3 : 1)) return dfail(d, MIJ_ERR_INVALID_ARG, "pitch smaller than a pixel row");
  DHIP(d, hipSetDevice(d->device));
  hipStream_t s = (hipStream_t)stream;
  d->last_stream = s;

  Geom g{};
  g.W = ps.W; g.H = ps.H; g.hs = ps.hs; g.vs = ps.vs; g.nl = ps.hs * ps.vs; g.bpm = g.nl + 2;
  g.mcux = (g.W + 8 * g.hs - 1) / (8 * g.hs); g.mcuy = (g.H + 8 * g.vs - 1) / (8 * g.vs);
  g.mcu_first = 0; g.mcu_count = (long long)g.mcux * g.mcuy; g.last_strip = 1;
  g.ri = ps.ri > 0 ? ps.ri : (int)std::min<long long>(g.mcu_count, 0x7FFFFFFF);   // no DRI: one interval = one lane (slow)
  const long long nseg = (g.mcu_count + g.ri - 1) / g.ri;
  const size_t scan_len = jpeg_bytes - ps.scan_off;
  const size_t nchunks = (scan_len + 16383) / 16384 + 1;
  const size_t ncoef = (size_t)g.mcu_count * g.bpm * 64;
  const size_t ysz = (size_t)g.mcux * g.hs * 8 * g.mcuy * g.vs * 8, csz = (size_t)g.mcux * 8 * g.mcuy * 8;
  if ((rc = ensure(d, d->d_scan, d->scan_cap, scan_len + 16)) || (rc = ensure(d, d->d_coef, d->coef_cap, ncoef)) ||
      (rc = ensure(d, d->d_planes, d->planes_cap, ysz + 2 * csz)) || (rc = ensure(d, d->d_seg_pos, d->seg_cap, (size_t)nseg + 1)))
    return rc;
  if (nchunks > d->chunk_cap) {
    (void)hipFree(d->d_chunk_cnt); (void)hipFree(d->d_chunk_base); d->chunk_cap = 0;
    DHIP(d, hipMalloc(&d->d_chunk_cnt, nchunks * sizeof(unsigned long long)));
    DHIP(d, hipMalloc(&d->d_chunk_base, nchunks * sizeof(unsigned long long)));
    d->chunk_cap = nchunks;
  }
  DHIP(d, hipEventRecord(d->ev0, s));
  // nvjpegDecodeJpegTransferToDevice (reference .cu:365): entropy-coded data + tables
  DHIP(d, hipMemcpyAsync(d->d_scan, jpeg + ps.scan_off, scan_len, hipMemcpyHostToDevice, s));
  DHIP(d, hipMemcpyAsync(d->d_tab, &ps.t, sizeof(DecTables), hipMemcpyHostToDevice, s));
  DHIP(d, hipMemsetAsync(d->d_flags, 0, 2 * sizeof(uint32_t), s));
  DHIP(d, hipStreamSynchronize(s));   // `ps` lives on this stack frame
  DHIP(d, launch_find_restarts(d->d_scan, scan_len, d->d_chunk_cnt, d->d_chunk_base, d->d_seg_pos, nseg, d->d_flags, d->d_res, s));
  DHIP(d, launch_huff_decode(g, d->d_scan, scan_len, d->d_seg_pos, nseg, d->d_tab, d->d_coef, d->d_flags + 1, s));
  uint8_t *py = d->d_planes, *pcb = py + ysz, *pcr = pcb + csz;
  DHIP(d, launch_idct(g, d->d_coef, d->d_tab, py, pcb, pcr, s));
  DHIP(d, launch_upsample_color(g, py, pcb, pcr, (uint8_t *)d_dst, pitch, plane_stride, output_format, s));
  DHIP(d, hipEventRecord(d->ev1, s));
  d->issued = true;
  return MIJ_OK;
}

int mij_decode_sync(mij_decoder *d, float *device_ms) {
  if (!d) return MIJ_ERR_INVALID_ARG;
  if (!d->issued) return dfail(d, MIJ_ERR_NOT_READY, "no decode has been issued on this handle");
  DHIP(d, hipSetDevice(d->device));
  DHIP(d, hipStreamSynchronize(d->last_stream));
  uint32_t flags[2] = {0, 0};
  DHIP(d, hipMemcpy(flags, d->d_flags, sizeof flags, hipMemcpyDeviceToHost));
  if (device_ms) { float t = 0; if (hipEventElapsedTime(&t, d->ev0, d->ev1) != hipSuccess) t = -1.f; *device_ms = t; }
  if (flags[1]) return dfail(d, MIJ_ERR_BAD_STREAM, "corrupt entropy-coded data");
  return MIJ_OK;
}

int mij_decode_host(mij_decoder *d, const uint8_t *jpeg, size_t jpeg_bytes, uint8_t *dst, size_t pitch, int output_format, int *width,
                    int *height) {
  if (!d || !jpeg || !dst) return dfail(d, MIJ_ERR_INVALID_ARG, "null argument");
  int w = 0, h = 0;
  int rc = mij_decode_info(jpeg, jpeg_bytes, &w, &h, nullptr, nullptr);
  if (rc) return dfail(d, rc, g_dec_err.c_str());
  const bool interleaved = output_format == MIJ_INPUT_BGRI || output_format == MIJ_INPUT_RGBI;
  const size_t row = (size_t)w * (interleaved ? 3 : 1);
  if (pitch < row) return dfail(d, MIJ_ERR_INVALID_ARG, "pitch smaller than a pixel row");
  const size_t bytes = interleaved ? row * h : row * h * 3;
  DHIP(d, hipSetDevice(d->device));
  if ((rc = ensure(d, d->d_out, d->out_cap, bytes))) return rc;
  rc = mij_decode_device(d, jpeg, jpeg_bytes, d->d_out, row, row * h, output_format, nullptr);
  if (rc) return rc;
  rc = mij_decode_sync(d, nullptr);
  if (rc) return rc;
  if (interleaved) DHIP(d, hipMemcpy2D(dst, pitch, d->d_out, row, row, (size_t)h, hipMemcpyDeviceToHost));
  else DHIP(d, hipMemcpy2D(dst, pitch, d->d_out, row, row, (size_t)h * 3, hipMemcpyDeviceToHost));
  if (width) *width = w;
  if (height) *height = h;
  return MIJ_OK;
}

int mij_residual_device(const void *d_a, const void *d_b, void *d_out, size_t n, int mode, void *stream) {
  if (!d_a || !d_b || !d_out) return MIJ_ERR_INVALID_ARG;
  hipError_t he = launch_residual((const uint8_t *)d_a, (const uint8_t *)d_b, (uint8_t *)d_out, n, mode, (hipStream_t)stream);
  return he == hipSuccess ? MIJ_OK : MIJ_ERR_HIP;
}

}  // extern "C"

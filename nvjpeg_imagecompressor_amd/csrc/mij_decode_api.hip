// mij_decode_api.hip -- host side of the decode half of the C ABI (include/mi_jpeg.h): marker parsing and table
// construction on the host (what nvjpegJpegStreamParse / nvjpegGetImageInfo do, reference ImageCompressorImpl.cu:335,362),
// everything else on the device (k_decode.inc). No CPU decode fallback.
#include "../../include/mi_jpeg.h"
#include "mij_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mij;

struct mij_decoder {
  int device = 0;
  std::string err;
  uint8_t *d_scan = nullptr; size_t scan_cap = 0;
  int16_t *d_coef = nullptr; size_t coef_cap = 0;
  uint8_t *d_planes = nullptr; size_t planes_cap = 0;
  DecTables *d_tab = nullptr;
  DecTables *d_tabs = nullptr; size_t tabs_cap = 0;   // generic route: one table snapshot per scan
  unsigned long long *d_seg_pos = nullptr; size_t seg_cap = 0;
  unsigned long long *d_chunk_cnt = nullptr, *d_chunk_base = nullptr; size_t chunk_cap = 0;
  uint32_t *d_flags = nullptr;    // [0] scratch for the scan kernel, [1] Huffman decode errors, [2] "a synchronisation pass changed a state"
  uint8_t *d_par_ws = nullptr; size_t par_ws_cap = 0;   // workspace of the parallel baseline decoder
  // generic route: independent scans run concurrently, each with its own restart-position workspace
  unsigned long long *d_scan_ws = nullptr; size_t scan_ws_cap = 0;
  unsigned long long *d_clean_len = nullptr;             // fast route: length of the clean stream
  uint8_t *d_clean = nullptr; size_t clean_cap = 0;      // un-stuffed copies of the scans that have no restart markers (k_decode_wave.inc)
  uint8_t *d_px_ws = nullptr; size_t px_ws_cap = 0;      // workspaces of the parallel progressive decoder (k_decode_prog.inc), one per scan
  uint32_t *d_px_flags = nullptr; size_t px_flags_cap = 0;   // PX_FLAG_WORDS words per scan
  std::vector<int> px_scans;                                  // last decode: kind of every scan the parallel decoder was tried on (0: not tried)
  hipStream_t aux[4]{};
  std::vector<hipEvent_t> scan_ev;
  hipEvent_t ev_ready{};
  bool aux_ok = false;
  int sync_passes = 0;
  DeviceResult *d_res = nullptr;
  uint8_t *d_out = nullptr; size_t out_cap = 0;
  uint8_t *d_out2 = nullptr; size_t out2_cap = 0;     // second layer of a secondary-compression pair
  hipStream_t last_stream = nullptr;
  hipEvent_t ev0{}, ev1{};
  bool ev_ok = false, issued = false;
  float last_ms = -1.f;
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

struct ScanInfo {
  ScanDesc sd{};
  size_t off = 0, len = 0;     // entropy-coded data of this scan inside the file
  DecTables tab{};             // the Huffman tables in force when the scan starts (DHT may be re-sent between scans)
};

struct Parsed {
  int W = 0, H = 0, hs = 0, vs = 0, ri = 0, ncomp = 0;
  bool progressive = false;
  int comp_id[3] = {0, 0, 0};
  bool have_sof = false;
  DecTables t{};
  bool have_tab[4] = {false, false, false, false};
  size_t scan_off = 0;         // first scan's data (fast path)
  std::vector<ScanInfo> scans;
  bool fast = false;           // one interleaved 3-component sequential scan: k_huff_decode takes it
};

// End of the entropy-coded data that starts at `i`: the next marker that is neither a stuffed FF00 nor RSTn.
static size_t scan_end(const uint8_t *p, size_t n, size_t i) {
  while (i + 1 < n) {
    const uint8_t *f = static_cast<const uint8_t *>(memchr(p + i, 0xFF, n - 1 - i));   // FF bytes are rare: let libc skip to them
    if (!f) return n;
    i = (size_t)(f - p);
    if (p[i + 1] != 0x00 && (p[i + 1] & 0xF8) != 0xD0 && p[i + 1] != 0xFF) return i;
    i++;
  }
  return n;
}

// 8-bit Huffman-coded frames: baseline / extended sequential (SOF0, SOF1) and progressive (SOF2); 1 component
// (greyscale) or 3 components with 1x1 chroma and luma h x v, h in {1,2,4}, v in {1,2}; Huffman and quantisation table
// ids 0..1 / 0..3. Anything else -> MIJ_ERR_BAD_STREAM with a reason.
// `total` != 0: p[0..n) is only a prefix of a file of `total` bytes (header parsing of a device-resident file); lengths that
// refer to the entropy-coded data are then taken from `total`.
static int parse_jpeg(const uint8_t *p, size_t n, Parsed &o, std::string &why, size_t total = 0) {
  if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) { why = "not a JPEG (no SOI)"; return MIJ_ERR_BAD_STREAM; }
  size_t i = 2;
  while (i + 4 <= n) {
    if (p[i] != 0xFF) { why = "marker expected"; return MIJ_ERR_BAD_STREAM; }
    const int m = p[i + 1];
    if (m == 0xFF) { i++; continue; }
    if (m == 0xD9) break;                                   // EOI
    if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) { i += 2; continue; }
    const size_t len = ((size_t)p[i + 2] << 8) | p[i + 3];
    if (len < 2 || i + 2 + len > n) { why = "truncated segment"; return MIJ_ERR_BAD_STREAM; }
    const uint8_t *s = p + i + 4;
    const size_t pl = len - 2;
    if (m == 0xDB) {
      size_t k = 0;
      while (k + 65 <= pl) {
        if ((s[k] >> 4) != 0 || (s[k] & 15) > 3) { why = "unsupported DQT (16-bit or id > 3)"; return MIJ_ERR_BAD_STREAM; }
        for (int z = 0; z < 64; z++) o.t.q[s[k] & 15][kZZ[z]] = s[k + 1 + z];
        k += 65;
      }
    } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
      // one frame per file (libjpeg: JERR_SOF_DUPLICATE). Every scan descriptor is frozen from the frame in force at its
      // SOS and the device buffers are sized from the frame: a second SOFn would let scans index outside them.
      if (o.have_sof) { why = "duplicate SOF marker"; return MIJ_ERR_BAD_STREAM; }
      o.have_sof = true;
      if (pl < 6 || s[0] != 8) { why = "only 8-bit samples"; return MIJ_ERR_BAD_STREAM; }
      o.progressive = m == 0xC2;
      o.ncomp = s[5];
      if ((o.ncomp != 1 && o.ncomp != 3) || pl < 6 + 3 * (size_t)o.ncomp) { why = "only 1- or 3-component frames"; return MIJ_ERR_BAD_STREAM; }
      o.H = (s[1] << 8) | s[2]; o.W = (s[3] << 8) | s[4];
      for (int c = 0; c < o.ncomp; c++) {
        const int hv = s[7 + 3 * c];
        o.comp_id[c] = s[6 + 3 * c];
        o.t.tq[c] = s[8 + 3 * c] & 3;
        if (c == 0) { o.hs = hv >> 4; o.vs = hv & 15; } else if (hv != 0x11) { why = "chroma sampling must be 1x1"; return MIJ_ERR_BAD_STREAM; }
      }
      if (o.ncomp == 1) { o.hs = o.vs = 1; o.t.tq[1] = o.t.tq[2] = o.t.tq[0]; }   // a lone component is never subsampled (A.2.2)
      if (o.W <= 0 || o.H <= 0) { why = "empty frame"; return MIJ_ERR_BAD_STREAM; }
      if (!((o.hs == 1 || o.hs == 2 || o.hs == 4) && (o.vs == 1 || o.vs == 2))) { why = "unsupported luma sampling"; return MIJ_ERR_BAD_STREAM; }
    } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
      why = "only Huffman-coded sequential (SOF0/SOF1) and progressive (SOF2) frames are handled";
      return MIJ_ERR_BAD_STREAM;
    } else if (m == 0xC4) {
      size_t k = 0;
      while (k + 17 <= pl) {
        const int tc = s[k] >> 4, th = s[k] & 15;
        if (tc > 1 || th > 1) { why = "unsupported DHT (table id > 1)"; return MIJ_ERR_BAD_STREAM; }
        const int t = th * 2 + tc;
        int cnt = 0;
        for (int l = 1; l <= 16; l++) cnt += s[k + l];
        if (cnt < 1 || cnt > 256 || k + 17 + (size_t)cnt > pl) { why = "bad DHT"; return MIJ_ERR_BAD_STREAM; }
        // The code counts must describe a prefix code (libjpeg jdhuff.c jpeg_make_d_derived_tbl: JERR_BAD_HUFF_TABLE): at
        // every length l the codes assigned so far must fit l bits. Otherwise `code << (9 - l)` below runs past look[]
        // and the device kernels would index vals[] out of range.
        { int c2 = 0; bool ok = true;
          for (int l = 1; l <= 16; l++) { c2 += s[k + l]; if (c2 > (1 << l)) ok = false; c2 <<= 1; }
          if (!ok) { why = "bad DHT (code counts are not a prefix code)"; return MIJ_ERR_BAD_STREAM; } }
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
      if (o.ncomp == 0) { why = "SOS before SOF"; return MIJ_ERR_BAD_STREAM; }
      const int ns = pl >= 1 ? s[0] : 0;
      if (ns < 1 || ns > o.ncomp || pl < 4 + 2 * (size_t)ns) { why = "bad SOS"; return MIJ_ERR_BAD_STREAM; }
      ScanInfo si;
      ScanDesc &sd = si.sd;
      sd.ncomp = ns;
      for (int c = 0; c < ns; c++) {
        int idx = -1;
        for (int q = 0; q < o.ncomp; q++) if (o.comp_id[q] == s[1 + 2 * c]) idx = q;
        if (idx < 0 || (c > 0 && idx <= sd.comp[c - 1])) { why = "bad component selector in SOS"; return MIJ_ERR_BAD_STREAM; }
        sd.comp[c] = idx;
        const int td = s[2 + 2 * c] >> 4, ta = s[2 + 2 * c] & 15;
        if (td > 1 || ta > 1) { why = "Huffman table id > 1 in SOS"; return MIJ_ERR_BAD_STREAM; }
        sd.td[c] = td * 2; sd.ta[c] = ta * 2 + 1;
      }
      sd.Ss = s[1 + 2 * ns]; sd.Se = s[2 + 2 * ns];
      const int Ah = s[3 + 2 * ns] >> 4;
      sd.Al = s[3 + 2 * ns] & 15;
      if (o.progressive) {
        if (sd.Ss > sd.Se || sd.Se > 63 || sd.Al > 13 || (sd.Ss == 0 && sd.Se != 0) || (sd.Ss > 0 && ns != 1) || (Ah != 0 && Ah != sd.Al + 1)) {
          why = "invalid progressive scan parameters"; return MIJ_ERR_BAD_STREAM;
        }
        sd.kind = sd.Ss == 0 ? (Ah == 0 ? 1 : 2) : (Ah == 0 ? 3 : 4);
      } else {
        if (sd.Ss != 0 || sd.Se != 63 || Ah != 0 || sd.Al != 0) { why = "invalid sequential scan parameters"; return MIJ_ERR_BAD_STREAM; }
        sd.kind = 0;
      }
      for (int c = 0; c < ns; c++) {
        if ((sd.kind <= 1) && !o.have_tab[sd.td[c]]) { why = "missing DC Huffman table"; return MIJ_ERR_BAD_STREAM; }
        if ((sd.kind == 0 || sd.kind >= 3) && !o.have_tab[sd.ta[c]]) { why = "missing AC Huffman table"; return MIJ_ERR_BAD_STREAM; }
      }
      const int mcux = (o.W + 8 * o.hs - 1) / (8 * o.hs), mcuy = (o.H + 8 * o.vs - 1) / (8 * o.vs);
      if (ns > 1) {
        if (ns != o.ncomp) { why = "partially interleaved scans are not handled"; return MIJ_ERR_BAD_STREAM; }
        sd.bw = mcux; sd.bh = mcuy; sd.nmcu = (long long)mcux * mcuy;
      } else if (o.ncomp == 1) {
        sd.bw = (o.W + 7) / 8; sd.bh = (o.H + 7) / 8; sd.nmcu = (long long)sd.bw * sd.bh;
      } else {
        const int cw = sd.comp[0] == 0 ? o.W : (o.W + o.hs - 1) / o.hs, ch = sd.comp[0] == 0 ? o.H : (o.H + o.vs - 1) / o.vs;
        sd.bw = (cw + 7) / 8; sd.bh = (ch + 7) / 8; sd.nmcu = (long long)sd.bw * sd.bh;
      }
      sd.ri = o.ri;
      si.off = i + 2 + len;
      // A sequential scan that carries every component is the only scan of its frame: its data runs to the end of the
      // file and need not be walked on the CPU (204 MB at the full size); anything else may be followed by more scans.
      si.len = (!o.progressive && ns == o.ncomp) ? (total ? total : n) - si.off : scan_end(p, n, si.off) - si.off;
      for (int c = 0; c < 3; c++) { o.t.td[c] = c < ns ? sd.td[c] / 2 : 0; o.t.ta[c] = c < ns ? sd.ta[c] / 2 : 0; }
      si.tab = o.t;
      if (o.scans.empty()) o.scan_off = si.off;
      o.scans.push_back(si);
      i = si.off + si.len;
      continue;
    }
    i += 2 + len;
  }
  if (o.scans.empty()) { why = "no SOS"; return MIJ_ERR_BAD_STREAM; }
  // later DQT segments may follow earlier scans: every scan's table snapshot gets the final quantisation tables
  for (auto &sc : o.scans) { memcpy(sc.tab.q, o.t.q, sizeof o.t.q); memcpy(sc.tab.tq, o.t.tq, sizeof o.t.tq); }
  o.fast = !o.progressive && o.scans.size() == 1 && o.ncomp == 3 && o.scans[0].sd.ncomp == 3;
  return MIJ_OK;
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
  (void)hipFree(d->d_res); (void)hipFree(d->d_out); (void)hipFree(d->d_out2); (void)hipFree(d->d_tabs); (void)hipFree(d->d_par_ws); (void)hipFree(d->d_scan_ws); (void)hipFree(d->d_clean); (void)hipFree(d->d_clean_len); (void)hipFree(d->d_px_ws); (void)hipFree(d->d_px_flags);
  if (d->aux_ok) { for (auto &q : d->aux) (void)hipStreamDestroy(q); (void)hipEventDestroy(d->ev_ready); }
  for (auto &v : d->scan_ev) (void)hipEventDestroy(v);
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
  if ((he = hipMalloc(&d->d_tab, sizeof(DecTables))) != hipSuccess || (he = hipMalloc(&d->d_flags, 4 * sizeof(uint32_t))) != hipSuccess ||
      (he = hipMalloc(&d->d_res, sizeof(DeviceResult))) != hipSuccess || (he = hipMemset(d->d_flags, 0, 4 * sizeof(uint32_t))) != hipSuccess ||
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
  DHIP(d, hipSetDevice(d->device));
  hipStream_t s = (hipStream_t)stream;
  // The file may already sit in device memory (e.g. straight out of this library's encoder, as in the difference-map
  // scheme): then only its header comes to the host for parsing and the kernels read the entropy-coded data in place.
  const uint8_t *d_file = nullptr;
  std::vector<uint8_t> hcopy;
  {
    hipPointerAttribute_t at{};
    if (hipPointerGetAttributes(&at, jpeg) == hipSuccess && at.type == hipMemoryTypeDevice) {
      d_file = jpeg;
      hcopy.resize(std::min<size_t>(jpeg_bytes, 1 << 16));
      DHIP(d, hipMemcpy(hcopy.data(), d_file, hcopy.size(), hipMemcpyDeviceToHost));
      jpeg = hcopy.data();
    } else (void)hipGetLastError();
  }
  Parsed ps; std::string why;
  int rc = parse_jpeg(jpeg, d_file ? hcopy.size() : jpeg_bytes, ps, why, d_file ? jpeg_bytes : 0);
  if (d_file && (rc || !ps.fast)) {
    // header longer than the prefix, or a multi-scan file whose scans must be located: bring the whole file over
    hcopy.resize(jpeg_bytes);
    DHIP(d, hipMemcpy(hcopy.data(), d_file, jpeg_bytes, hipMemcpyDeviceToHost));
    jpeg = hcopy.data();
    d_file = nullptr;
    ps = Parsed(); why.clear();
    rc = parse_jpeg(jpeg, jpeg_bytes, ps, why, 0);
  }
  if (rc) return dfail(d, rc, why.c_str());
  if (pitch < (size_t)ps.W * (interleaved ? 3 : 1)) return dfail(d, MIJ_ERR_INVALID_ARG, "pitch smaller than a pixel row");
  d->last_stream = s;

  Geom g{};
  g.W = ps.W; g.H = ps.H; g.hs = ps.hs; g.vs = ps.vs; g.nl = ps.hs * ps.vs; g.bpm = g.nl + 2;
  g.mcux = (g.W + 8 * g.hs - 1) / (8 * g.hs); g.mcuy = (g.H + 8 * g.vs - 1) / (8 * g.vs);
  g.mcu_first = 0; g.mcu_count = (long long)g.mcux * g.mcuy; g.last_strip = 1;
  g.ri = ps.ri > 0 ? ps.ri : (int)std::min<long long>(g.mcu_count, 0x7FFFFFFF);   // no DRI: one interval = one lane (slow)
  geom_finish(g);
  const size_t ncoef = (size_t)g.mcu_count * g.bpm * 64;
  const size_t ysz = (size_t)g.mcux * g.hs * 8 * g.mcuy * g.vs * 8, csz = (size_t)g.mcux * 8 * g.mcuy * 8;
  // entropy-coded bytes of all scans, from the first scan's data to the end of the last one's
  const size_t data_off = ps.scans.front().off;
  const size_t scan_len = ps.fast ? jpeg_bytes - data_off : ps.scans.back().off + ps.scans.back().len - data_off;
  // The subsequence-parallel and the wave decoders keep positions inside a scan as 32-bit byte offsets (k_decode_par.inc
  // ParReader, k_decode_wave.inc): a file with 4 GiB or more of entropy-coded data (65535 x 65535 noise at q100 can get
  // there) is refused instead of being decoded from truncated offsets.
  if (scan_len >= ((size_t)1 << 32) - 64) return dfail(d, MIJ_ERR_BAD_STREAM, "entropy-coded data of 4 GiB or more is not supported");
  long long max_seg = 1;
  size_t max_len = 0;
  for (const auto &sc : ps.scans) {
    const long long ns = sc.sd.ri > 0 ? (sc.sd.nmcu + sc.sd.ri - 1) / sc.sd.ri : 1;
    max_seg = std::max(max_seg, ns);
    max_len = std::max(max_len, sc.len);
  }
  if (ps.fast) { max_seg = (g.mcu_count + g.ri - 1) / g.ri; max_len = scan_len; }
  const size_t nchunks = 2 * ((max_len + 16383) / 16384 + 1);      // (the fast route keeps two counts per chunk)
  if ((rc = ensure(d, d->d_scan, d->scan_cap, scan_len + 64 /* BitReader window slack */)) || (rc = ensure(d, d->d_coef, d->coef_cap, ncoef)) ||
      (rc = ensure(d, d->d_planes, d->planes_cap, ysz + 2 * csz)) || (rc = ensure(d, d->d_seg_pos, d->seg_cap, (size_t)max_seg + 1)) ||
      (rc = ensure(d, d->d_tabs, d->tabs_cap, ps.scans.size())))
    return rc;
  if (nchunks > d->chunk_cap) {
    (void)hipFree(d->d_chunk_cnt); (void)hipFree(d->d_chunk_base); d->chunk_cap = 0;
    DHIP(d, hipMalloc(&d->d_chunk_cnt, nchunks * sizeof(unsigned long long)));
    DHIP(d, hipMalloc(&d->d_chunk_base, nchunks * sizeof(unsigned long long)));
    d->chunk_cap = nchunks;
  }
  DHIP(d, hipEventRecord(d->ev0, s));
  // nvjpegDecodeJpegTransferToDevice (reference .cu:365): entropy-coded data + tables
  const uint8_t *scan_dev = d->d_scan;
  if (d_file) scan_dev = d_file + data_off;       // device-resident file: decode in place
  else DHIP(d, hipMemcpyAsync(d->d_scan, jpeg + data_off, scan_len, hipMemcpyHostToDevice, s));
  DHIP(d, hipMemsetAsync(d->d_flags, 0, 2 * sizeof(uint32_t), s));
  const DecTables *d_final = d->d_tab;
  DcFix dc_fix{};                          // set by the parallel route: the IDCT completes the DC terms
  if (ps.fast) {
    d->px_scans.clear();                // mij_decode_px_report speaks of THIS decode: no progressive scans in it
    DHIP(d, hipMemcpyAsync(d->d_tab, &ps.t, sizeof(DecTables), hipMemcpyHostToDevice, s));
    DHIP(d, hipStreamSynchronize(s));   // `ps` lives on this stack frame
    static const bool lanes_only = getenv("MIJ_DECODE_LANES") != nullptr;   // A/B switch: one lane per restart interval (k_huff_decode)
    if (lanes_only) {
      DHIP(d, launch_find_restarts(scan_dev, scan_len, d->d_chunk_cnt, d->d_chunk_base, d->d_seg_pos, max_seg, d->d_flags, d->d_res, s));
      DHIP(d, launch_huff_decode(g, scan_dev, scan_len, d->d_seg_pos, max_seg, d->d_tab, d->d_coef, d->d_flags + 1, s));
    } else {
      // subsequence-parallel decode (k_decode_par.inc) of an un-stuffed, marker-free copy of the scan; its synchronisation
      // passes are checked from the host, so this call waits for them (the IDCT / colour kernels that follow are still
      // asynchronous)
      if ((rc = ensure(d, d->d_par_ws, d->par_ws_cap, par_workspace_bytes(scan_len, max_seg))) || (rc = ensure(d, d->d_clean, d->clean_cap, scan_len + 256)))
        return rc;
      if (!d->d_clean_len) DHIP(d, hipMalloc(&d->d_clean_len, sizeof(unsigned long long)));
      DHIP(d, launch_clean_scan(scan_dev, scan_len, d->d_chunk_cnt, d->d_chunk_base, d->d_seg_pos, max_seg, d->d_clean, d->d_clean_len, d->d_flags,
                                d->d_res, s));
      DHIP(d, launch_par_decode(g, d->d_clean, scan_len, d->d_clean_len, d->d_seg_pos, max_seg, d->d_tab, d->d_coef, d->d_par_ws, d->d_flags + 2,
                                d->d_flags + 1, &d->sync_passes, s, &dc_fix));
      if (d->sync_passes < 0) {
        dc_fix = DcFix{};  // the states did not settle within 64 passes (adversarial data): exact lane-per-interval decode of the raw stream
        DHIP(d, launch_find_restarts(scan_dev, scan_len, d->d_chunk_cnt, d->d_chunk_base, d->d_seg_pos, max_seg, d->d_flags, d->d_res, s));
        DHIP(d, launch_huff_decode(g, scan_dev, scan_len, d->d_seg_pos, max_seg, d->d_tab, d->d_coef, d->d_flags + 1, s));
      }
    }
  } else {
    // generic route (k_decode_scans.inc): scans in file order into a zeroed coefficient buffer
    std::vector<DecTables> tabs(ps.scans.size());
    for (size_t i = 0; i < ps.scans.size(); i++) tabs[i] = ps.scans[i].tab;
    DHIP(d, hipMemcpyAsync(d->d_tabs, tabs.data(), tabs.size() * sizeof(DecTables), hipMemcpyHostToDevice, s));
    DHIP(d, hipMemsetAsync(d->d_coef, 0, ncoef * sizeof(int16_t), s));
    DHIP(d, hipStreamSynchronize(s));   // `tabs` lives on this stack frame
    // Scans that touch disjoint coefficients (different components, or disjoint spectral bands of one component) are
    // independent: each runs on one of four streams after the earlier scans it overlaps with. A refinement scan thus
    // waits for the first scan of its band, while e.g. the chroma scans proceed next to the luma ones.
    if (!d->aux_ok) {
      for (auto &q : d->aux) DHIP(d, hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
      DHIP(d, hipEventCreateWithFlags(&d->ev_ready, hipEventDisableTiming));
      d->aux_ok = true;
    }
    while (d->scan_ev.size() < ps.scans.size()) {
      hipEvent_t ev;
      DHIP(d, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      d->scan_ev.push_back(ev);
    }
    // per-scan workspace: restart positions (ns + 1) and two chunk-count arrays
    std::vector<size_t> o_seg(ps.scans.size()), o_cnt(ps.scans.size()), o_base(ps.scans.size()), o_len(ps.scans.size()), o_clean(ps.scans.size());
    size_t words = 0, clean_bytes = 0;
    static const bool lanes_only = getenv("MIJ_DECODE_SCAN_LANES") != nullptr;   // A/B switch: the one-lane walk for scans without restart markers
    for (size_t i = 0; i < ps.scans.size(); i++) {
      const ScanInfo &sc = ps.scans[i];
      const size_t ns = sc.sd.ri > 0 ? (size_t)((sc.sd.nmcu + sc.sd.ri - 1) / sc.sd.ri) : 1, nch = (sc.len + 16383) / 16384 + 1;
      o_seg[i] = words; words += ns + 1;
      o_cnt[i] = words; words += nch;
      o_base[i] = words; words += nch + 1;
      o_len[i] = words; words += 1;
      o_clean[i] = clean_bytes;
      if (sc.sd.ri == 0 && !lanes_only) clean_bytes += (sc.len + 256 + 255) & ~(size_t)255;
    }
    if ((rc = ensure(d, d->d_scan_ws, d->scan_ws_cap, words))) return rc;
    if (clean_bytes && (rc = ensure(d, d->d_clean, d->clean_cap, clean_bytes))) return rc;
    // progressive scans without restart markers: the parallel decoder (k_decode_prog.inc) goes first, each scan with a workspace
    // of its own (scans run concurrently); the wave decoder is its fallback, decided on the device
    std::vector<size_t> o_px(ps.scans.size(), (size_t)-1);
    size_t px_bytes = 0;
    for (size_t i = 0; i < ps.scans.size(); i++) {
      const ScanInfo &sc = ps.scans[i];
      if (ps.progressive && sc.sd.ri == 0 && !lanes_only && px_supported(sc.sd)) {
        o_px[i] = px_bytes;
        px_bytes += (px_workspace_bytes(sc.sd, g, sc.len) + 511) & ~(size_t)255;
      }
    }
    // The parallel decoder's workspace is sized by the FILE (hypothesis lists: 16 bytes x 2 x at least 8 M per AC refinement scan, ~2.5 GB for a
    // luma refinement of 8320x40000), held by the handle, four scans' worth at a time. It must never be the reason a decode fails: above a
    // byte budget (MIJ_PX_WS_BUDGET_MB, default 24,576 = 24 GiB of the 288) or when the allocation is refused, the largest scans give theirs up
    // and are walked by the wave decoder (exact, needs none of it, slower), until the rest fits.
    {
      static const size_t budget = []() { const char *e = getenv("MIJ_PX_WS_BUDGET_MB"); const long long mb = e ? atoll(e) : 24576; return (size_t)(mb < 0 ? 0 : mb) << 20; }();
      auto recount = [&]() {
        px_bytes = 0;
        for (size_t i = 0; i < ps.scans.size(); i++)
          if (o_px[i] != (size_t)-1) { o_px[i] = px_bytes; px_bytes += (px_workspace_bytes(ps.scans[i].sd, g, ps.scans[i].len) + 511) & ~(size_t)255; }
      };
      auto drop_largest = [&]() -> bool {
        size_t big = (size_t)-1, big_bytes = 0;
        for (size_t i = 0; i < ps.scans.size(); i++)
          if (o_px[i] != (size_t)-1) { const size_t b = px_workspace_bytes(ps.scans[i].sd, g, ps.scans[i].len); if (b >= big_bytes) { big = i; big_bytes = b; } }
        if (big == (size_t)-1) return false;
        o_px[big] = (size_t)-1;
        recount();
        return true;
      };
      while (px_bytes > budget && drop_largest()) {}
      while (px_bytes) {
        if (px_bytes <= d->px_ws_cap) break;
        (void)hipFree(d->d_px_ws); d->d_px_ws = nullptr; d->px_ws_cap = 0;
        if (hipMalloc(&d->d_px_ws, px_bytes) == hipSuccess) { d->px_ws_cap = px_bytes; break; }
        (void)hipGetLastError();          // refused: not an error of this decode
        if (!drop_largest()) break;
      }
      if (px_bytes && (rc = ensure(d, d->d_px_flags, d->px_flags_cap, 4 * PX_FLAG_WORDS * ps.scans.size()))) return rc;
    }
    DHIP(d, hipEventRecord(d->ev_ready, s));
    for (auto &q : d->aux) DHIP(d, hipStreamWaitEvent(q, d->ev_ready, 0));
    for (size_t i = 0; i < ps.scans.size(); i++) {
      const ScanInfo &sc = ps.scans[i];
      hipStream_t q = d->aux[i & 3];
      for (size_t j = 0; j < i; j++) {
        const ScanDesc &a = ps.scans[j].sd, &b = sc.sd;
        bool comp = false;
        for (int x = 0; x < a.ncomp; x++) for (int y = 0; y < b.ncomp; y++) comp |= a.comp[x] == b.comp[y];
        if (comp && a.Ss <= b.Se && b.Ss <= a.Se && (j & 3) != (i & 3)) DHIP(d, hipStreamWaitEvent(q, d->scan_ev[j], 0));
      }
      {   // backstop: the scan's block grid must lie inside the frame the coefficient buffer was sized for
        const ScanDesc &b = sc.sd;
        const int gw = b.ncomp > 1 ? g.mcux : (ps.ncomp == 1 || b.comp[0] != 0 ? g.mcux : g.mcux * g.hs);
        const int gh = b.ncomp > 1 ? g.mcuy : (ps.ncomp == 1 || b.comp[0] != 0 ? g.mcuy : g.mcuy * g.vs);
        bool ok = b.bw >= 1 && b.bh >= 1 && b.bw <= gw && b.bh <= gh && b.nmcu == (long long)b.bw * b.bh && b.ncomp >= 1 && b.ncomp <= ps.ncomp;
        for (int c = 0; c < b.ncomp && ok; c++) ok = b.comp[c] >= 0 && b.comp[c] < ps.ncomp;
        if (!ok) return dfail(d, MIJ_ERR_BAD_STREAM, "scan geometry does not match the frame");
      }
      const uint8_t *base = d->d_scan + (sc.off - data_off);
      const long long ns = sc.sd.ri > 0 ? (sc.sd.nmcu + sc.sd.ri - 1) / sc.sd.ri : 1;
      unsigned long long *seg = d->d_scan_ws + o_seg[i];
      if (sc.sd.ri == 0 && !lanes_only) {
        // no restart markers: the scan is one chain of symbols -- a whole wave walks it (k_decode_wave.inc)
        bool same_dc = true;         // a DC scan whose components share one Huffman table parses the same whatever the block-in-MCU
        for (int c = 1; c < sc.sd.ncomp; c++)
          same_dc = same_dc && (sc.sd.td[c] == sc.sd.td[0] || memcmp(sc.tab.look[sc.sd.td[c]], sc.tab.look[sc.sd.td[0]], sizeof sc.tab.look[0]) == 0);
        const bool px = o_px[i] != (size_t)-1;
        DHIP(d, launch_scan_decode_wave(g, sc.sd, base, sc.len, d->d_scan_ws + o_cnt[i], d->d_scan_ws + o_base[i], d->d_scan_ws + o_len[i],
                                        d->d_clean + o_clean[i], d->d_tabs + i, d->d_coef, d->d_flags, d->d_res, d->d_flags + 1, q,
                                        px ? d->d_px_ws + o_px[i] : nullptr, px ? d->d_px_flags + PX_FLAG_WORDS * i : nullptr, same_dc));
      } else {
        if (sc.sd.ri > 0) DHIP(d, launch_find_restarts(base, sc.len, d->d_scan_ws + o_cnt[i], d->d_scan_ws + o_base[i], seg, ns, d->d_flags, d->d_res, q));
        else DHIP(d, hipMemsetAsync(seg, 0, sizeof(unsigned long long), q));
        DHIP(d, launch_scan_decode(g, sc.sd, base, sc.len, seg, ns, d->d_tabs + i, d->d_coef, d->d_flags + 1, q));
      }
      DHIP(d, hipEventRecord(d->scan_ev[i], q));
    }
    for (size_t i = 0; i < ps.scans.size(); i++) DHIP(d, hipStreamWaitEvent(s, d->scan_ev[i], 0));
    d_final = d->d_tabs + (ps.scans.size() - 1);
    if (px_bytes) {
      // which scans did the parallel progressive decoder take? (mij_decode_px_report; MIJ_PX_DEBUG=1 prints it)
      d->px_scans.assign(ps.scans.size(), 0);
      for (size_t i = 0; i < ps.scans.size(); i++) d->px_scans[i] = o_px[i] != (size_t)-1 ? ps.scans[i].sd.kind : 0;
      static const bool dbg = getenv("MIJ_PX_DEBUG") != nullptr;
      if (dbg) {
        DHIP(d, hipStreamSynchronize(s));
        std::vector<uint32_t> f(PX_FLAG_WORDS * ps.scans.size());
        DHIP(d, hipMemcpy(f.data(), d->d_px_flags, f.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < ps.scans.size(); i++) {
          if (!d->px_scans[i]) continue;
          const uint32_t *fi = f.data() + PX_FLAG_WORDS * i;
          fprintf(stderr, "[px] scan %zu kind %d Ss %d Se %d Al %d bytes %zu: %s (unresolved anchors %u, anchors found wrong %d)\n", i, ps.scans[i].sd.kind, ps.scans[i].sd.Ss,
                  ps.scans[i].sd.Se, ps.scans[i].sd.Al, ps.scans[i].len, fi[0] ? "FELL BACK to the wave decoder" : "parallel", fi[2], (int)fi[3]);
          if (d->px_scans[i] == 4 && fi[4]) fprintf(stderr, "[px]        candidate lists written %u, anchors the chain decided among candidates %u, adopted %u\n", fi[4], fi[5], fi[8]);
        }
      }
    } else d->px_scans.clear();
  }
  uint8_t *py = d->d_planes, *pcb = py + ysz, *pcr = pcb + csz;
  if (idct_color_supported(g, output_format)) DHIP(d, launch_idct_color(g, d->d_coef, d_final, nullptr, dc_fix, (uint8_t *)d_dst, pitch, output_format, s));
  else {
    DHIP(d, launch_idct(g, d->d_coef, d_final, py, pcb, pcr, s, dc_fix));
    DHIP(d, launch_upsample_color(g, py, pcb, pcr, (uint8_t *)d_dst, pitch, plane_stride, output_format, s));
  }
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
  { float t = 0; if (hipEventElapsedTime(&t, d->ev0, d->ev1) != hipSuccess) t = -1.f; d->last_ms = t; }
  if (device_ms) *device_ms = d->last_ms;
  if (flags[1]) return dfail(d, MIJ_ERR_BAD_STREAM, "corrupt entropy-coded data");
  return MIJ_OK;
}

int mij_decode_last_ms(const mij_decoder *d, float *device_ms) {
  if (!d || !device_ms) return MIJ_ERR_INVALID_ARG;
  if (d->last_ms < 0) return MIJ_ERR_NOT_READY;
  *device_ms = d->last_ms;
  return MIJ_OK;
}

int mij_decoder_device(const mij_decoder *d) { return d ? d->device : -1; }

int mij_decode_px_report(mij_decoder *d, int *scans_tried, int *scans_parallel) {
  if (!d) return MIJ_ERR_INVALID_ARG;
  if (!d->issued) return dfail(d, MIJ_ERR_NOT_READY, "no decode has been issued on this handle");
  DHIP(d, hipSetDevice(d->device));
  DHIP(d, hipStreamSynchronize(d->last_stream));
  int tried = 0, par = 0;
  if (!d->px_scans.empty()) {
    std::vector<uint32_t> f(PX_FLAG_WORDS * d->px_scans.size());
    DHIP(d, hipMemcpy(f.data(), d->d_px_flags, f.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < d->px_scans.size(); i++)
      if (d->px_scans[i]) { tried++; if (!f[PX_FLAG_WORDS * i]) par++; }
  }
  if (scans_tried) *scans_tried = tried;
  if (scans_parallel) *scans_parallel = par;
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

int mij_secondary_decode_host(mij_decoder *d, const uint8_t *primary, size_t primary_bytes, const uint8_t *secondary, size_t secondary_bytes,
                              uint8_t *dst, size_t pitch, int output_format, int *width, int *height) {
  return mij_secondary_decode_host_ex(d, nullptr, primary, primary_bytes, secondary, secondary_bytes, dst, pitch, output_format, width, height);
}

int mij_secondary_decode_host_ex(mij_decoder *d, const mij_secondary_params *sp, const uint8_t *primary, size_t primary_bytes,
                                 const uint8_t *secondary, size_t secondary_bytes, uint8_t *dst, size_t pitch, int output_format, int *width,
                                 int *height) {
  if (!d || !primary || !secondary || !dst) return dfail(d, MIJ_ERR_INVALID_ARG, "null argument");
  int gain_shift = 0;
  if (sp) {       // only the gain matters on the way back: quality and sampling of each layer are in its file
    if (sp->struct_size != sizeof(mij_secondary_params)) return dfail(d, MIJ_ERR_INVALID_ARG, "mij_secondary_params.struct_size matches no known layout");
    gain_shift = sp->gain <= 1 ? 0 : sp->gain == 2 ? 1 : sp->gain == 4 ? 2 : sp->gain == 8 ? 3 : -1;
    if (gain_shift < 0) return dfail(d, MIJ_ERR_INVALID_ARG, "mij_secondary_params.gain must be 0 (= 1), 1, 2, 4 or 8");
  }
  int w = 0, h = 0, w2 = 0, h2 = 0;
  int rc = mij_decode_info(primary, primary_bytes, &w, &h, nullptr, nullptr);
  if (!rc) rc = mij_decode_info(secondary, secondary_bytes, &w2, &h2, nullptr, nullptr);
  if (rc) return dfail(d, rc, g_dec_err.c_str());
  if (w != w2 || h != h2) return dfail(d, MIJ_ERR_BAD_STREAM, "the two layers have different dimensions");
  const bool interleaved = output_format == MIJ_INPUT_BGRI || output_format == MIJ_INPUT_RGBI;
  if (!interleaved && output_format != MIJ_INPUT_BGR && output_format != MIJ_INPUT_RGB) return dfail(d, MIJ_ERR_INVALID_ARG, "unknown output format");
  const size_t row = (size_t)w * (interleaved ? 3 : 1);
  if (pitch < row) return dfail(d, MIJ_ERR_INVALID_ARG, "pitch smaller than a pixel row");
  const size_t bytes = row * h * (interleaved ? 1 : 3);
  DHIP(d, hipSetDevice(d->device));
  if ((rc = ensure(d, d->d_out, d->out_cap, bytes)) || (rc = ensure(d, d->d_out2, d->out2_cap, bytes))) return rc;
  rc = mij_decode_device(d, primary, primary_bytes, d->d_out, row, row * h, output_format, nullptr);
  if (!rc) rc = mij_decode_sync(d, nullptr);
  if (!rc) rc = mij_decode_device(d, secondary, secondary_bytes, d->d_out2, row, row * h, output_format, nullptr);
  if (!rc) rc = mij_decode_sync(d, nullptr);
  if (rc) return rc;
  if (launch_residual(d->d_out, d->d_out2, d->d_out, bytes, +1, nullptr, gain_shift) != hipSuccess) return dfail(d, MIJ_ERR_HIP, "residual kernel");   // I' = clip(D + (R' - 128) / gain)
  DHIP(d, hipMemcpy2D(dst, pitch, d->d_out, row, row, (size_t)h * (interleaved ? 1 : 3), hipMemcpyDeviceToHost));
  if (width) *width = w;
  if (height) *height = h;
  return MIJ_OK;
}

int mij_residual_device(const void *d_a, const void *d_b, void *d_out, size_t n, int mode, void *stream) {
  return mij_residual_gain_device(d_a, d_b, d_out, n, mode, 1, stream);
}

int mij_residual_gain_device(const void *d_a, const void *d_b, void *d_out, size_t n, int mode, int gain, void *stream) {
  if (!d_a || !d_b || !d_out) return MIJ_ERR_INVALID_ARG;
  const int sh = gain <= 1 ? 0 : gain == 2 ? 1 : gain == 4 ? 2 : gain == 8 ? 3 : -1;
  if (sh < 0) return MIJ_ERR_INVALID_ARG;
  hipError_t he = launch_residual((const uint8_t *)d_a, (const uint8_t *)d_b, (uint8_t *)d_out, n, mode, (hipStream_t)stream, sh);
  return he == hipSuccess ? MIJ_OK : MIJ_ERR_HIP;
}

}  // extern "C"

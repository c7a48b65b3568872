// mij_api.hip -- host side of the C ABI declared in include/mi_jpeg.h.
// Owns the device workspace (the reference's "env", ImageCompressorImpl.cu:19-65) and sequences the kernels of
// mij_kernels.hip on the caller's stream. There is no CPU fallback: without a HIP device every entry point fails.
#include "../../include/mi_jpeg.h"
#include "mij_internal.h"

#include <algorithm>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mij;

struct mij_encoder {
  mij_encoder_params p{};
  Geom g{};
  std::string err;
  Quant hq{};
  Quant *d_qt = nullptr;
  DeviceTables *d_tab = nullptr;
  uint32_t *d_hist_own = nullptr, *d_hist = nullptr;
  int16_t *d_coef = nullptr, *d_dc = nullptr;
  size_t coef_count = 0, coef_alloc = 0;   // coefficients of the strip; bytes of d_coef (whole tiles)
  uint8_t *d_scratch = nullptr;
  size_t slot_bytes = 0;
  // progressive output: the ten scans run concurrently, each with its own workspace (allocated only when asked for)
  struct ProgScan {
    ScanDesc sd{};
    int Ah = 0;
    long long nseg = 0;
    long long seg0 = 0;               // strips: index, in the whole image's scan, of the strip's first restart interval (RSTn numbering)
    size_t slot = 0;
    uint8_t *scratch = nullptr;
    uint32_t *seg_bytes = nullptr, *seg_ff = nullptr, *hist = nullptr, *ovf = nullptr;
    uint8_t *flag = nullptr;          // per interval: 1 = the lane-per-block coder left it to the serial kernel
    bool fast = false;                // scan coded by k_encode_prog2.inc
    bool narrow = false;              // refinement scan: emit with 16-word strips (dropped when too many intervals overflow them)
    uint32_t seen_recoded = 0;
    int stream_index = 0;
    unsigned long long *seg_off = nullptr, *chunk_total = nullptr, *chunk_base = nullptr;
    DeviceTables *tab = nullptr;
    DeviceResult *res = nullptr;
  } ps[10];
  uint8_t *d_prog = nullptr;          // one allocation behind all of the above
  uint32_t *d_prog_hist = nullptr;    // the ten scans' statistics, contiguous (10 x 4 x 257 words): what a sharded encode all-reduces in one collective
  bool prog_stats_done = false, prog_emitted = false;   // strip protocol: mij_encode_prog_statistics / _emit issued for the current image
  hipStream_t prog_stream[4]{};
  hipEvent_t prog_ev[5]{};
  bool prog_streams = false;
  DeviceTables *h_prog_tab = nullptr; // pinned: the ten tables + results come back in one go
  DeviceResult *h_prog_res = nullptr;
  long long nseg = 0;
  uint32_t *d_seg_bytes = nullptr, *d_seg_ff = nullptr;
  unsigned long long *d_seg_off = nullptr, *d_chunk_total = nullptr, *d_chunk_base = nullptr;
  uint32_t *d_ovf = nullptr;
  unsigned long long *d_status = nullptr;   // fused entropy coder: look-back status words + ticket
  uint32_t *d_redo = nullptr;               // set by the fused coder when its result is unusable
  bool fuse = false, fused_run = false;
  uint8_t *d_out = nullptr;
  bool out_uncached = false;   // d_out came from mij_encoder_reserve_output as device-uncached memory (peers write into it)
  size_t capacity = 0;  // scan-data capacity (bytes after HDR_AREA)
  DeviceResult *d_res = nullptr, *h_res = nullptr, *h_res_dev = nullptr;   // h_res_dev: the device's address of the page-locked h_res
  uint8_t *d_src = nullptr;
  size_t d_src_bytes = 0;
  uint8_t *h_out = nullptr;
  size_t h_out_cap = 0;
  uint8_t *d_sec = nullptr;   // secondary compression: decoded first layer and residual, same layout as d_src
  size_t d_sec_bytes = 0;
  mij_encoder *sec_enc = nullptr;   // second layer coded with its own quality / sampling (mij_secondary_encode_host_ex): created on
  int sec_quality = 0, sec_css = 0; // first use, kept while the parameters stay the same
  hipStream_t s_copy = nullptr, s_work = nullptr;   // mij_encode_host: upload stream / kernel stream
  hipEvent_t ev_chunk[2]{};                          // "chunk i has landed" (ping-pong)
  bool host_streams = false;
  hipEvent_t ev[8]{};
  hipEvent_t ev_xdone{}, ev_tab{};   // transform complete / tables built (mij_encode_tables on another stream)
  hipEvent_t ev_place[3]{};          // mij_encode_place with timing on: before K6, between K6 and k_put, behind k_put
  bool ev_place_ok = false, place_timed = false, place_put = false;
  bool tables_early = false;         // this image's tables were built by mij_encode_tables
  hipEvent_t ev_done{};      // recorded behind the result copy: mij_encode_result waits for THIS encode only, so that a caller
                             // who alternates two handles on one stream keeps the GPU busy while it collects a result
  bool ev_ok = false, timing = false, timed_run = false;
  float ms[MIJ_NUM_STAGE_TIMES]{};
  bool hist_clean[2] = {true, true};                        // own statistics buffers known to be zero (cleared at creation / by k_build_tables)
  bool collected = true;                                    // mij_encode_result has waited for the handle's last image
  bool placed_as_root = false;                              // the last mij_encode_place assembled the file in this handle's buffer
  bool k4_narrow = false;                                   // fast entropy coder with 16-word strips (5 waves per SIMD); see note_recoded
  uint32_t seen_recoded = 0;
  bool dc_folded = false;                                  // this image's DC statistics were taken inside k_transform
  bool no_dc_fold = getenv("MIJ_NO_DC_FOLD") != nullptr;   // A/B switch: always use k_dc_stats
  bool transformed = false, issued = false, static_tables_ready = false, wait_event = false, sharded_pending = false;
  hipStream_t last_stream = nullptr;
};

static thread_local std::string g_create_err;

static int fail(mij_encoder *e, int code, const char *what, hipError_t he = hipSuccess) {
  std::string m = what;
  if (he != hipSuccess) { m += ": "; m += hipGetErrorString(he); }
  if (e) e->err = m; else g_create_err = m;
  fprintf(stderr, "[ERROR] mi_jpeg: %s\n", m.c_str());
  return code;
}
#define HIPCHK(e, x)                                                        \
  do {                                                                      \
    hipError_t he_ = (x);                                                   \
    if (he_ != hipSuccess) return fail((e), MIJ_ERR_HIP, #x, he_);          \
  } while (0)

static const uint8_t kStdLumQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57,
                                     69, 56, 14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64,
                                     81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kStdChrQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99,
                                     99, 99, 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                     99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

// IJG quality scaling (nvjpegEncoderParamsSetQuality semantics, reference ImageCompressorImpl.cu:30).
static void make_quant(int quality, Quant &q) {
  quality = std::min(100, std::max(1, quality));
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 64; i++) {
      long v = ((long)(t ? kStdChrQ[i] : kStdLumQ[i]) * scale + 50L) / 100L;
      v = std::min(255L, std::max(1L, v));
      q.q[t][i] = (uint16_t)v;
      const float d = (float)(8 * v);
      q.recip[t][i] = (1.0f / d) * (1.0f + 0x1p-17f);   // see quant_magic() in k_common.inc
    }
}

static int css_factors(int css, int &hs, int &vs) {
  switch (css) {
    case MIJ_CSS_444: hs = 1; vs = 1; return 0;
    case MIJ_CSS_422: hs = 2; vs = 1; return 0;
    case MIJ_CSS_420: hs = 2; vs = 2; return 0;
    case MIJ_CSS_440: hs = 1; vs = 2; return 0;
    case MIJ_CSS_411: hs = 4; vs = 1; return 0;
    case MIJ_CSS_410: hs = 4; vs = 2; return 0;
    default: return -1;
  }
}

// Restart interval = unit of GPU parallelism (one wavefront each) and of strip sharding. The entropy coder works through an
// interval in batches of 64 blocks, so AUTO picks an interval whose block count is a multiple of 64 -- every batch full --
// and about 256 blocks long: short enough that an eighth of the headline image still holds more intervals (5,078) than the
// chip has wave slots (4,096), long enough that the per-interval work (table set-up, padding, marker) stays small.
// Measured at the full size, 4:2:2 (MCUs = blocks per interval / ms per image): 104 = 416 / 1.446 (round 1's choice: a
// divisor of the MCUs per row, its last batch half empty), 64 = 256 / 1.405, 96 = 384 / 1.415, 128 = 512 / 1.411,
// 192 = 768 / 1.447. An interval need not divide the MCU row: strips of a sharded encode are then cut where interval and
// row boundaries coincide (sharded.rows_per_restart_unit). Progressive output has its own rule below.
static int choose_restart_interval(int mcux, int bpm, bool progressive = false, long long luma_blocks = 0) {
  (void)mcux;
  if (progressive) {
    // Progressive output is whole-image only, and the lane-per-block coder (k_encode_prog2.inc) works through a
    // single-component scan's interval in batches of 64 blocks too. How long: about two intervals of the luma scans per wave
    // slot of the chip. Measured at the full size: 104 -> 5.9 ms, 448...896 -> 4.9-5.0, 1024 -> 5.1, 1536 -> 5.3.
    const long long k = std::min(16LL, std::max(1LL, (luma_blocks + 64 * 4096) / (64 * 8192)));
    return (int)(64 * k);
  }
  int gcd = 64, b = bpm;
  while (b) { const int t = gcd % b; gcd = b; b = t; }
  const int unit = 64 / gcd;                              // smallest interval whose block count is a multiple of 64
  const int k = std::max(1, (256 + unit * bpm / 2) / (unit * bpm));
  return unit * k;
}

extern "C" {

const char *mij_version(void) { return "mi_jpeg 0.2 (gfx950)"; }
int mij_abi_version(void) { return MIJ_ABI_VERSION; }
#ifndef MIJ_SOURCE_HASH
#define MIJ_SOURCE_HASH "unknown"
#endif
const char *mij_source_hash(void) { return MIJ_SOURCE_HASH; }

int mij_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char *mij_last_error(const mij_encoder *enc) { return enc ? enc->err.c_str() : g_create_err.c_str(); }

void mij_encoder_destroy(mij_encoder *e) {
  if (e && e->sec_enc) { mij_encoder_destroy(e->sec_enc); e->sec_enc = nullptr; }
  if (!e) return;
  (void)hipSetDevice(e->p.device);
  if (e->last_stream || e->issued) (void)hipStreamSynchronize(e->last_stream);
  (void)hipFree(e->d_qt); (void)hipFree(e->d_tab); (void)hipFree(e->d_hist_own); (void)hipFree(e->d_coef); (void)hipFree(e->d_dc);
  (void)hipFree(e->d_scratch); (void)hipFree(e->d_seg_bytes); (void)hipFree(e->d_seg_ff); (void)hipFree(e->d_seg_off); (void)hipFree(e->d_chunk_total); (void)hipFree(e->d_chunk_base); (void)hipFree(e->d_ovf); (void)hipFree(e->d_status); (void)hipFree(e->d_redo);
  (void)hipFree(e->d_out); (void)hipFree(e->d_res); (void)hipFree(e->d_src); (void)hipFree(e->d_sec);
  if (e->h_res) (void)hipHostFree(e->h_res);
  if (e->h_out) (void)hipHostFree(e->h_out);
  if (e->ev_place_ok) for (auto &v : e->ev_place) (void)hipEventDestroy(v);
  if (e->ev_ok) { for (auto &v : e->ev) (void)hipEventDestroy(v); (void)hipEventDestroy(e->ev_done); (void)hipEventDestroy(e->ev_xdone); (void)hipEventDestroy(e->ev_tab); }
  (void)hipFree(e->d_prog);
  if (e->h_prog_tab) (void)hipHostFree(e->h_prog_tab);
  if (e->h_prog_res) (void)hipHostFree(e->h_prog_res);
  if (e->prog_streams) {
    for (auto &q : e->prog_stream) (void)hipStreamDestroy(q);
    for (auto &v : e->prog_ev) (void)hipEventDestroy(v);
  }
  if (e->host_streams) {
    (void)hipStreamDestroy(e->s_copy); (void)hipStreamDestroy(e->s_work);
    for (auto &v : e->ev_chunk) (void)hipEventDestroy(v);
  }
  delete e;
}

// Validates the parameters and derives the geometry (pure arithmetic: no device needed). `p` receives the normalised copy.
static int derive_geometry(const mij_encoder_params *p_in, mij_encoder_params &pcopy, Geom &g);

int mij_geometry_query(const mij_encoder_params *p_in, mij_geometry *o) {
  if (!p_in || !o) return fail(nullptr, MIJ_ERR_INVALID_ARG, "null argument");
  mij_encoder e;            // only p / g are used
  int rc = derive_geometry(p_in, e.p, e.g);
  if (rc) return rc;
  return mij_encoder_geometry(&e, o);
}

static int derive_geometry(const mij_encoder_params *p_in, mij_encoder_params &pcopy, Geom &g) {
  // The caller's struct may be older (shorter) than this library's: copy only what it has, zero the rest, and refuse sizes
  // that match no layout this library knows (a caller that never set struct_size lands here too: width is not a size).
  constexpr size_t kV1 = offsetof(mij_encoder_params, progressive);       // ABI 1 + struct_size: no `progressive`
  if (p_in->struct_size != sizeof(mij_encoder_params) && p_in->struct_size != kV1)
    return fail(nullptr, MIJ_ERR_INVALID_ARG, "mij_encoder_params.struct_size matches no known layout (set it to sizeof(mij_encoder_params))");
  pcopy = mij_encoder_params{};
  memcpy(&pcopy, p_in, p_in->struct_size);
  pcopy.struct_size = (uint32_t)sizeof(mij_encoder_params);
  const mij_encoder_params *p = &pcopy;
  int hs, vs;
  if (p->width <= 0 || p->height <= 0 || p->width > 65535 || p->height > 65535)
    return fail(nullptr, MIJ_ERR_INVALID_ARG, "width/height must be in 1..65535");
  if (p->quality < 1 || p->quality > 100) return fail(nullptr, MIJ_ERR_INVALID_ARG, "quality must be in 1..100");
  if (css_factors(p->css, hs, vs)) return fail(nullptr, MIJ_ERR_INVALID_ARG, "unsupported chroma subsampling");
  g = Geom{};
  g.W = p->width; g.H = p->height; g.hs = hs; g.vs = vs; g.nl = hs * vs; g.bpm = g.nl + 2;
  g.mcux = (g.W + 8 * hs - 1) / (8 * hs);
  g.mcuy = (g.H + 8 * vs - 1) / (8 * vs);
  g.wib0 = (g.W + 7) / 8; g.hib0 = (g.H + 7) / 8;
  g.crows = (g.H + vs - 1) / vs;
  g.quality = p->quality;
  int ri = p->restart_interval;
  if (ri == MIJ_RESTART_AUTO) ri = choose_restart_interval(g.mcux, g.bpm, p->progressive != 0, (long long)g.wib0 * g.hib0);
  if (ri < 1 || ri > 65535) return fail(nullptr, MIJ_ERR_INVALID_ARG, "restart_interval must be 1..65535 MCUs (or MIJ_RESTART_AUTO)");
  g.ri = ri;
  int row0 = p->strip_mcu_row0, rows = p->strip_mcu_rows;
  if (rows == 0) { row0 = 0; rows = g.mcuy; }
  if (row0 < 0 || rows < 0 || row0 + rows > g.mcuy) return fail(nullptr, MIJ_ERR_INVALID_ARG, "strip outside the image");
  g.mcu_first = (long long)row0 * g.mcux;
  g.mcu_count = (long long)rows * g.mcux;
  if (g.mcu_first % ri) return fail(nullptr, MIJ_ERR_INVALID_ARG, "strip does not start on a restart-interval boundary");
  g.last_strip = (row0 + rows == g.mcuy);
  if (!g.last_strip && (g.mcu_count % ri)) return fail(nullptr, MIJ_ERR_INVALID_ARG, "strip does not end on a restart-interval boundary");
  g.y_origin = row0 * 8 * vs;
  geom_finish(g);
  if (p->progressive && (row0 != 0 || rows != g.mcuy)) {
    // Progressive output in strips (round 5): a scan of ONE component counts its restart intervals in blocks of that component, in raster
    // order over the component -- so a strip of MCU rows is a whole number of intervals in every scan only if the interval divides the
    // blocks per block row of luma (mcux * hs), of chroma and the MCUs per row (mcux): ri | mcux, and the width a whole number of MCUs.
    if (g.wib0 != g.mcux * hs || (g.mcux % ri) != 0)
      return fail(nullptr, MIJ_ERR_INVALID_ARG, "progressive output in strips needs a width of whole MCUs and a restart interval that divides the MCUs per row");
  }
  return MIJ_OK;
}

int mij_encoder_create(const mij_encoder_params *p_in, mij_encoder **out) {
  if (!p_in || !out) return fail(nullptr, MIJ_ERR_INVALID_ARG, "null argument");
  *out = nullptr;
  mij_encoder_params pcopy;
  Geom g0;
  int rc0 = derive_geometry(p_in, pcopy, g0);
  if (rc0) return rc0;
  const mij_encoder_params *p = &pcopy;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, MIJ_ERR_NO_DEVICE, "no HIP device: mi_jpeg has no CPU fallback");
  if (p->device < 0 || p->device >= ndev) return fail(nullptr, MIJ_ERR_INVALID_ARG, "device ordinal out of range");
  HIPCHK(nullptr, hipSetDevice(p->device));

  mij_encoder *e = new (std::nothrow) mij_encoder();
  if (!e) return fail(nullptr, MIJ_ERR_ALLOC, "out of host memory");
  e->p = *p;
  e->g = g0;
  Geom &g = e->g;
  const int ri = g.ri, hs = g.hs, vs = g.vs;
  e->nseg = (g.mcu_count + ri - 1) / ri;
  e->coef_count = (size_t)g.mcu_count * g.bpm * 64;
  e->slot_bytes = (((size_t)ri * g.bpm * MAX_BLOCK_BYTES + 8) + 255) & ~(size_t)255;
  e->capacity = e->coef_count + 65536;
  make_quant(p->quality, e->hq);

#define CRCHK(x) do { hipError_t he_ = (x); if (he_ != hipSuccess) { int rc_ = fail(nullptr, MIJ_ERR_HIP, #x, he_); mij_encoder_destroy(e); return rc_; } } while (0)
  CRCHK(hipMalloc(&e->d_qt, sizeof(Quant)));
  CRCHK(hipMemcpy(e->d_qt, &e->hq, sizeof(Quant), hipMemcpyHostToDevice));
  CRCHK(hipMalloc(&e->d_tab, sizeof(DeviceTables)));
  CRCHK(hipMalloc(&e->d_hist_own, 2 * 4 * 257 * sizeof(uint32_t)));       // two buffers, used in turn (k_build_tables clears the other one)
  CRCHK(hipMemset(e->d_hist_own, 0, 2 * 4 * 257 * sizeof(uint32_t)));
  e->d_hist = e->d_hist_own;
  e->coef_alloc = (size_t)coef_tiles(g, g.mcu_count) * coef_tile_bytes(g);      // whole tiles (mij_internal.h: coefficient layout)
  CRCHK(hipMalloc(&e->d_coef, e->coef_alloc));
  CRCHK(hipMalloc(&e->d_dc, (e->coef_count / 64) * sizeof(int16_t)));
  const size_t seg_alloc = (size_t)e->nseg, scratch_alloc = e->slot_bytes * (size_t)e->nseg;
  CRCHK(hipMalloc(&e->d_scratch, scratch_alloc));
  CRCHK(hipMalloc(&e->d_seg_bytes, seg_alloc * sizeof(uint32_t)));
  CRCHK(hipMalloc(&e->d_seg_ff, seg_alloc * sizeof(uint32_t)));
  CRCHK(hipMalloc(&e->d_seg_off, seg_alloc * sizeof(unsigned long long)));
  { const size_t nch = (seg_alloc + 1023) / 1024;
    CRCHK(hipMalloc(&e->d_chunk_total, nch * sizeof(unsigned long long)));
    CRCHK(hipMalloc(&e->d_chunk_base, nch * sizeof(unsigned long long))); }
  CRCHK(hipMalloc(&e->d_ovf, sizeof(uint32_t)));
  CRCHK(hipMemset(e->d_ovf, 0, sizeof(uint32_t)));
  CRCHK(hipMalloc(&e->d_status, (seg_alloc + 1) * sizeof(unsigned long long)));
  CRCHK(hipMalloc(&e->d_redo, sizeof(uint32_t)));
  CRCHK(hipMemset(e->d_redo, 0, sizeof(uint32_t)));
  // Opt-in experiment (MIJ_FUSE=1): K4 with the size scan and the stuffing + compaction folded in (decoupled look-back).
  // Measured SLOWER than the three separate kernels (0.91 vs 0.59 + 0.01 + 0.10 ms): the placement work is a
  // latency-bound chain per interval that K6 runs at 32 waves per CU and the fused kernel at 16, on top of the coder.
  e->fuse = getenv("MIJ_FUSE") != nullptr;
  CRCHK(hipMalloc(&e->d_out, HDR_AREA + e->capacity + 64));
  CRCHK(hipMalloc(&e->d_res, sizeof(DeviceResult)));
  CRCHK(hipMemset(e->d_res, 0, sizeof(DeviceResult)));
  e->k4_narrow = p->quality <= 97 && getenv("MIJ_K4_WIDE") == nullptr;
  CRCHK(hipHostMalloc(&e->h_res, sizeof(DeviceResult), hipHostMallocDefault));
  { void *dp = nullptr; if (hipHostGetDevicePointer(&dp, e->h_res, 0) == hipSuccess) e->h_res_dev = (DeviceResult *)dp; else (void)hipGetLastError(); }
  memset(e->h_res, 0, sizeof(DeviceResult));
  if (p->progressive) {
    // libjpeg's jpeg_simple_progression for YCbCr (jcparam.c); a single-component scan has one block per "MCU"
    static const int script[10][6] = {{3, 0, 0, 0, 0, 1}, {1, 0, 1, 5, 0, 2}, {1, 2, 1, 63, 0, 1}, {1, 1, 1, 63, 0, 1}, {1, 0, 6, 63, 0, 2},
                                      {1, 0, 1, 63, 2, 1}, {3, 0, 0, 0, 1, 0}, {1, 2, 1, 63, 1, 0}, {1, 1, 1, 63, 1, 0}, {1, 0, 1, 63, 1, 0}};
    size_t total = 0;
    auto take = [&](size_t bytes) { const size_t o = total; total += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_scr[10], o_sb[10], o_sf[10], o_so[10], o_ct[10], o_cb[10], o_h[10], o_t[10], o_r[10], o_v[10], o_f[10];
    for (int i = 0; i < 10; i++) {
      mij_encoder::ProgScan &q = e->ps[i];
      ScanDesc &sd = q.sd;
      sd.ncomp = script[i][0];
      sd.comp[0] = script[i][1]; sd.comp[1] = 1; sd.comp[2] = 2;
      if (sd.ncomp == 3) sd.comp[0] = 0;
      sd.Ss = script[i][2]; sd.Se = script[i][3]; q.Ah = script[i][4]; sd.Al = script[i][5];
      sd.kind = sd.Ss == 0 ? (q.Ah == 0 ? 1 : 2) : (q.Ah == 0 ? 3 : 4);
      sd.ri = ri;
      // (a strip: the scan's block rows that lie in the strip's MCU rows; the coefficient buffer and every index below are the strip's)
      const int srow0 = (int)(g.mcu_first / g.mcux), srows = (int)(g.mcu_count / g.mcux);
      if (sd.ncomp > 1) { sd.bw = g.mcux; sd.bh = srows; q.slot = e->slot_bytes; q.seg0 = g.mcu_first / ri; }
      else {
        const int c = sd.comp[0];
        const int cw = c == 0 ? g.W : (g.W + hs - 1) / hs, ch = c == 0 ? g.H : (g.H + vs - 1) / vs;
        const int per = c == 0 ? vs : 1;                     // block rows of this component per MCU row
        const int bh_all = (ch + 7) / 8;
        sd.bw = (cw + 7) / 8;
        sd.bh = std::max(0, std::min(bh_all, (srow0 + srows) * per) - srow0 * per);
        q.seg0 = ((long long)srow0 * per * sd.bw) / ri;
        q.slot = (((size_t)ri * MAX_BLOCK_BYTES + 8) + 255) & ~(size_t)255;
      }
      sd.nmcu = (long long)sd.bw * sd.bh;
      q.nseg = (sd.nmcu + ri - 1) / ri;
      const size_t nch = (size_t)((q.nseg + 1023) / 1024);
      o_scr[i] = take(q.slot * (size_t)q.nseg); o_sb[i] = take((size_t)q.nseg * 4); o_sf[i] = take((size_t)q.nseg * 4);
      o_so[i] = take((size_t)q.nseg * 8); o_ct[i] = take(nch * 8); o_cb[i] = take(nch * 8); o_h[i] = 0;
      o_t[i] = take(sizeof(DeviceTables)); o_r[i] = take(sizeof(DeviceResult)); o_v[i] = take(4); o_f[i] = take((size_t)q.nseg);
      q.fast = prog2_supported(sd) && getenv("MIJ_PROG_SERIAL") == nullptr;   // A/B switch: the lane-per-interval kernel only
      // first guess; the count of overflowed intervals corrects it after the first image (the last luma refinement of a
      // high-quality file has blocks of more than 512 bits: q95 synthetic, a third of its intervals)
      q.narrow = q.fast && sd.kind == 4 && (sd.comp[0] != 0 || sd.Al > 0 || p->quality <= 85) && getenv("MIJ_PROG_WIDE") == nullptr;
    }
    const size_t o_hist = take(10 * 4 * 257 * 4);            // contiguous: one collective reduces all ten over the ranks
    CRCHK(hipMalloc(&e->d_prog, total));
    e->d_prog_hist = (uint32_t *)(e->d_prog + o_hist);
    for (int i = 0; i < 10; i++) {
      o_h[i] = o_hist + (size_t)i * 4 * 257 * 4;
      mij_encoder::ProgScan &q = e->ps[i];
      q.scratch = e->d_prog + o_scr[i]; q.seg_bytes = (uint32_t *)(e->d_prog + o_sb[i]); q.seg_ff = (uint32_t *)(e->d_prog + o_sf[i]);
      q.seg_off = (unsigned long long *)(e->d_prog + o_so[i]); q.chunk_total = (unsigned long long *)(e->d_prog + o_ct[i]);
      q.chunk_base = (unsigned long long *)(e->d_prog + o_cb[i]); q.hist = (uint32_t *)(e->d_prog + o_h[i]);
      q.tab = (DeviceTables *)(e->d_prog + o_t[i]); q.res = (DeviceResult *)(e->d_prog + o_r[i]); q.ovf = (uint32_t *)(e->d_prog + o_v[i]);
      q.flag = e->d_prog + o_f[i];
      CRCHK(hipMemset(q.flag, 0, (size_t)q.nseg));
      CRCHK(hipMemset(q.ovf, 0, 4));
      CRCHK(hipMemset(q.res, 0, sizeof(DeviceResult)));
    }
    CRCHK(hipHostMalloc(&e->h_prog_tab, 10 * sizeof(DeviceTables), hipHostMallocDefault));
    CRCHK(hipHostMalloc(&e->h_prog_res, 10 * sizeof(DeviceResult), hipHostMallocDefault));
    for (auto &st : e->prog_stream) CRCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &v : e->prog_ev) CRCHK(hipEventCreateWithFlags(&v, hipEventDisableTiming));
    e->prog_streams = true;
  }
  for (auto &v : e->ev) CRCHK(hipEventCreate(&v));
  CRCHK(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
  CRCHK(hipEventCreateWithFlags(&e->ev_xdone, hipEventDisableTiming));
  CRCHK(hipEventCreateWithFlags(&e->ev_tab, hipEventDisableTiming));
  e->ev_ok = true;
#undef CRCHK
  *out = e;
  return MIJ_OK;
}

int mij_encoder_geometry(const mij_encoder *e, mij_geometry *o) {
  if (!e || !o) return MIJ_ERR_INVALID_ARG;
  const Geom &g = e->g;
  o->hs = g.hs; o->vs = g.vs; o->mcu_w = 8 * g.hs; o->mcu_h = 8 * g.vs;
  o->mcus_per_row = g.mcux; o->mcu_rows = g.mcuy; o->blocks_per_mcu = g.bpm; o->restart_interval = g.ri;
  o->strip_first_mcu = g.mcu_first; o->strip_mcus = g.mcu_count;
  o->strip_y0 = g.y_origin;
  const int y1 = std::min(g.H, (int)((g.mcu_first + g.mcu_count) / g.mcux) * 8 * g.vs);
  o->strip_rows = y1 - g.y_origin;
  return MIJ_OK;
}

int mij_encoder_enable_timing(mij_encoder *e, int on) {
  if (!e) return MIJ_ERR_INVALID_ARG;
  e->timing = on != 0;
  return MIJ_OK;
}

int mij_set_histogram_buffer(mij_encoder *e, uint32_t *d_hist) {
  if (!e) return MIJ_ERR_INVALID_ARG;
  e->d_hist = d_hist ? d_hist : e->d_hist_own;
  e->hist_clean[0] = e->hist_clean[1] = false;
  return MIJ_OK;
}

int mij_histogram_device(mij_encoder *e, uint32_t **d_hist, size_t *count) {
  if (!e || !d_hist) return MIJ_ERR_INVALID_ARG;
  *d_hist = e->d_hist;
  if (count) *count = 4 * 257;
  return MIJ_OK;
}

// Stage A (+ fused statistics) over MCU rows [row0, row0 + rows) of this handle's strip. The first range of an image
// resets the statistics, the last one adds the DC statistics; mij_encode_host streams an image through in several
// ranges while the next one is still being uploaded.
static int transform_rows(mij_encoder *e, const void *d_src, size_t pitch, size_t plane_stride, int fmt, hipStream_t s,
                          int row0, int rows, bool first, bool last) {
  const bool interleaved = fmt == MIJ_INPUT_RGBI || fmt == MIJ_INPUT_BGRI;
  const Geom &g = e->g;
  if (first) {
    e->last_stream = s;
    e->timed_run = e->timing;
    e->transformed = false;
    e->tables_early = false;
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[0], s));
    if (e->p.optimized_huffman && !e->p.progressive) {
      // The handle's own statistics buffers alternate, and the table kernel of one image clears the buffer of the next; a
      // caller-owned buffer (sharded path: a tensor the collective reduces in place) is cleared here.
      // (Only when the handle's previous image has been collected: its table kernel is then known to have finished, on
      // whatever stream it ran.)
      const bool own = e->d_hist == e->d_hist_own || e->d_hist == e->d_hist_own + 4 * 257;
      int c = e->d_hist == e->d_hist_own ? 0 : 1;
      if (own && e->collected && !e->hist_clean[c] && e->hist_clean[c ^ 1]) { c ^= 1; e->d_hist = e->d_hist_own + c * 4 * 257; }
      if (!(own && e->collected && e->hist_clean[c])) HIPCHK(e, hipMemsetAsync(e->d_hist, 0, 4 * 257 * sizeof(uint32_t), s));
      if (own) e->hist_clean[c] = false;
    }
  }
  TransformArgs a{};
  a.src = (const uint8_t *)d_src; a.pitch = pitch; a.plane_stride = plane_stride;
  const bool rgb_order = fmt == MIJ_INPUT_RGB || fmt == MIJ_INPUT_RGBI;
  const int kR[3] = {19595, -11059, 32768}, kB[3] = {7471, 32768, -5329};
  for (int i = 0; i < 3; i++) { a.fA[i] = (rgb_order ? kR[i] : kB[i]) / 65536.0f; a.fC[i] = (rgb_order ? kB[i] : kR[i]) / 65536.0f; }
  Geom sub = g;
  const long long skip = (long long)row0 * g.mcux;
  sub.mcu_first = g.mcu_first + skip;
  sub.mcu_count = std::min((long long)rows * g.mcux, g.mcu_count - skip);
  a.coef = e->d_coef; a.dc = e->d_dc; a.range_skip = skip;   // tiles and the compact DC array count from the strip's first MCU
  a.recip_dev = &e->d_qt->recip[0][0];
  a.hist = (e->p.optimized_huffman && !e->p.progressive) ? e->d_hist : nullptr;   // progressive gathers per scan instead
  a.write_dc = e->p.progressive ? 1 : 0;
  // The DC statistics come out of the same kernel when no DC prediction crosses one of its tiles (256 luma blocks): the whole
  // strip in this one call, every tile starting a restart interval (AUTO's intervals do), no dummy blocks whose DC is
  // patched afterwards. Otherwise k_dc_stats takes them from the compact DC array once the last range is through.
  const int nl_ = g.hs * g.vs, mpt = 256 / nl_ * (nl_ == 8 ? 4 : nl_ == 4 ? 2 : 1);    // MCUs a workgroup takes per pass (TCfg::SMPT)
  const bool dummies = g.mcux * g.hs > g.wib0 || g.mcuy * g.vs > g.hib0;
  if (first) e->dc_folded = a.hist && last && !dummies && mpt % g.ri == 0 && sub.mcu_first % g.ri == 0 && !e->no_dc_fold;
  a.fold_dc = e->dc_folded ? 1 : 0;
  if (sub.mcu_count > 0) HIPCHK(e, launch_transform(sub, a, interleaved ? 1 : 0, s));
  if (last) {
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[1], s));
    if (e->p.optimized_huffman && !e->p.progressive && !e->dc_folded) HIPCHK(e, launch_dc_stats(g, e->d_dc, e->d_hist, s));
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[2], s));
    HIPCHK(e, hipEventRecord(e->ev_xdone, s));
    e->transformed = true;
  }
  return MIJ_OK;
}

static int check_input(mij_encoder *e, const void *src, size_t pitch, int fmt) {
  if (!e || !src) return fail(e, MIJ_ERR_INVALID_ARG, "null argument");
  const bool interleaved = fmt == MIJ_INPUT_RGBI || fmt == MIJ_INPUT_BGRI;
  if (!interleaved && fmt != MIJ_INPUT_RGB && fmt != MIJ_INPUT_BGR) return fail(e, MIJ_ERR_INVALID_ARG, "unknown input format");
  if (pitch < (size_t)e->g.W * (interleaved ? 3 : 1)) return fail(e, MIJ_ERR_INVALID_ARG, "pitch smaller than a pixel row");
  return MIJ_OK;
}

// The fast entropy coder comes in two strip sizes (k_encode.inc): a handle stays on the narrow, faster one until more than
// 1 % of an image's restart intervals overflowed it and went through the roomy coder.
static void note_recoded(mij_encoder *e) {
  const uint32_t d = e->h_res->recoded - e->seen_recoded;
  e->seen_recoded = e->h_res->recoded;
  if (e->k4_narrow && (long long)d * 100 > e->nseg) e->k4_narrow = false;
}

static int strip_mcu_rows(const Geom &g) { return (int)((g.mcu_count + g.mcux - 1) / g.mcux); }

int mij_encode_transform(mij_encoder *e, const void *d_src, size_t pitch, size_t plane_stride, int fmt, void *stream) {
  int rc = check_input(e, d_src, pitch, fmt);
  if (rc) return rc;
  HIPCHK(e, hipSetDevice(e->p.device));
  return transform_rows(e, d_src, pitch, plane_stride, fmt, (hipStream_t)stream, 0, strip_mcu_rows(e->g), true, true);
}

static int run_tail(mij_encoder *e, hipStream_t s, bool tables) {
  const Geom &g = e->g;
  uint32_t *other = nullptr;
  if (tables && e->p.optimized_huffman && !e->p.progressive) {
    if (e->d_hist == e->d_hist_own) { other = e->d_hist_own + 4 * 257; e->hist_clean[1] = true; }
    else if (e->d_hist == e->d_hist_own + 4 * 257) { other = e->d_hist_own; e->hist_clean[0] = true; }
  }
  if (tables) HIPCHK(e, launch_build_tables(g, e->d_hist, e->p.optimized_huffman ? 1 : 0, e->d_qt, e->d_tab, e->d_out, e->d_res, s, other));
  return MIJ_OK;
}

// Progressive output (k_encode_prog.inc): the ten scans of libjpeg's jpeg_simple_progression. Only the final placement
// of a scan depends on the scans before it, so all ten are coded concurrently on four streams, each into its own
// workspace: gather statistics -> K3 builds the table(s) -> emit -> K5 sizes; one synchronisation brings the tables and
// sizes to the host, which writes DHT + SOS for every scan at its final offset and launches the ten compactions (K6)
// concurrently. Returns with the file complete.
// ---- progressive output: pieces shared by the whole-image path (encode_progressive) and the strip protocol below -----------
// The ten scans only READ the coefficients, so any of them may run beside any other. They go to the streams longest first, each to
// the stream with the least work queued (cost ~ blocks x band width, refinement scans twice that: the two Y refinement scans are a
// third of all the work and must not share a stream, which the file order i & 3 made them do).
static void prog_plan(mij_encoder *e, int (&order)[10], int (&owner)[10]) {
  double cost[10], load[4] = {0, 0, 0, 0};
  for (int i = 0; i < 10; i++) {
    const ScanDesc &sd = e->ps[i].sd;
    order[i] = i;
    cost[i] = (double)sd.nmcu * (sd.kind <= 2 ? 1.5 * (sd.kind == 1 ? 2 : 1) : (sd.Se - sd.Ss + 1) * (sd.kind == 4 ? 2.0 : 1.0));
  }
  static const int nstreams = getenv("MIJ_PROG_STREAMS") ? std::min(4, std::max(1, atoi(getenv("MIJ_PROG_STREAMS")))) : 3;   // experiment knobs; measured at the full size:
  // 1 stream 7.06 ms, 2: 6.16, 3: 5.87, 4: 6.56 (longest first) / 6.40, 6.04, 6.00 (file order): the scan kernels nearly fill the chip alone
  static const int lpt = getenv("MIJ_PROG_ORDER") ? atoi(getenv("MIJ_PROG_ORDER")) : 1;
  if (lpt) std::sort(order, order + 10, [&](int x, int y) { return cost[x] > cost[y]; });
  for (int k = 0; k < 10; k++) {
    int best = 0;
    for (int q4 = 1; q4 < nstreams; q4++) if (load[q4] < load[best]) best = q4;
    if (!lpt) best = order[k] % nstreams;
    owner[order[k]] = best; load[best] += cost[order[k]];
  }
}
// statistics of scan i into its slice of d_prog_hist (DC refinement scans have none)
static int prog_gather(mij_encoder *e, int i, hipStream_t st) {
  static const uint32_t one = 1;
  const Geom &g = e->g;
  mij_encoder::ProgScan &q = e->ps[i];
  if (q.sd.kind == 2) return MIJ_OK;
  HIPCHK(e, hipMemsetAsync(q.hist, 0, 4 * 257 * sizeof(uint32_t), st));
  // (K3 builds all four tables; the ones this scan does not use get a single count so that they are well formed:
  //  the lane-per-block gather kernels write it themselves, the serial one gets it by copy)
  if (q.fast) HIPCHK(e, launch_prog2(g, q.sd, 1, e->d_coef, e->d_dc, q.tab, q.scratch, q.slot, q.seg_bytes, q.seg_ff, q.hist, q.flag, q.nseg, st, false, nullptr));
  else {
    HIPCHK(e, launch_prog_encode(g, q.sd, 1, e->d_coef, q.tab, q.scratch, q.slot, q.seg_bytes, q.seg_ff, q.hist, q.nseg, st));
    for (int w = 0; w < 4; w++) {
      const bool used = q.sd.kind == 1 ? (w == 0 || w == 2) : (w == (q.sd.comp[0] ? 3 : 1));
      if (!used) HIPCHK(e, hipMemcpyAsync(q.hist + w * 257, &one, sizeof one, hipMemcpyHostToDevice, st));
    }
  }
  return MIJ_OK;
}
// scan i: its table(s) from the statistics (to the host as well: the DHT segment), its restart intervals' bits, their sizes
static int prog_tables_emit(mij_encoder *e, int i, hipStream_t st) {
  const Geom &g = e->g;
  mij_encoder::ProgScan &q = e->ps[i];
  if (q.sd.kind != 2) {
    // (K3 also writes a baseline header into the first HDR_AREA bytes of d_out; the progressive file starts after them)
    HIPCHK(e, launch_build_tables(g, q.hist, 1, e->d_qt, q.tab, e->d_out, q.res, st));
    HIPCHK(e, hipMemcpyAsync(&e->h_prog_tab[i], q.tab, sizeof(DeviceTables), hipMemcpyDeviceToHost, st));
  }
  if (q.fast) {
    // lane per block; intervals it hands back (forced flushes of jcphuff.c, oversized blocks) go through the serial kernel
    HIPCHK(e, launch_prog2(g, q.sd, 0, e->d_coef, e->d_dc, q.tab, q.scratch, q.slot, q.seg_bytes, q.seg_ff, q.hist, q.flag, q.nseg, st, q.narrow, q.res));
    if (q.sd.kind >= 3) HIPCHK(e, launch_prog_encode(g, q.sd, 0, e->d_coef, q.tab, q.scratch, q.slot, q.seg_bytes, q.seg_ff, q.hist, q.nseg, st, q.flag));
  } else HIPCHK(e, launch_prog_encode(g, q.sd, 0, e->d_coef, q.tab, q.scratch, q.slot, q.seg_bytes, q.seg_ff, q.hist, q.nseg, st));
  HIPCHK(e, launch_scan(q.seg_bytes, q.seg_ff, q.seg_off, q.nseg, q.chunk_total, q.chunk_base, q.ovf, q.res, st));
  HIPCHK(e, hipMemcpyAsync(&e->h_prog_res[i], q.res, sizeof(DeviceResult), hipMemcpyDeviceToHost, st));
  return MIJ_OK;
}
static void prog_note_recoded(mij_encoder *e) {      // as note_recoded for K4: more than 1 % of the intervals re-coded -> wide strips from now on
  for (int i = 0; i < 10; i++) {
    mij_encoder::ProgScan &q = e->ps[i];
    const uint32_t d = e->h_prog_res[i].recoded - q.seen_recoded;
    q.seen_recoded = e->h_prog_res[i].recoded;
    if (q.narrow && (long long)d * 100 > q.nseg) q.narrow = false;
  }
}
// what precedes scan i's entropy-coded data in the file (jcmarker.c: frame header in front of the first scan; per scan the DHT of the
// tables it uses, DRI before the first SOS, SOS). Depends on the image-wide statistics only: every rank of a sharded encode builds the same.
static void prog_header(const mij_encoder *e, int i, std::vector<uint8_t> &h) {
  static const uint8_t zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
                                 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61,
                                 54, 47, 55, 62, 63};
  const Geom &g = e->g;
  const mij_encoder::ProgScan &q = e->ps[i];
  h.clear();
  auto put = [&](int b) { h.push_back((uint8_t)b); };
  auto put16 = [&](int v) { put(v >> 8); put(v & 255); };
  if (i == 0) {
    put16(0xFFD8);
    put16(0xFFE0); put16(16); put('J'); put('F'); put('I'); put('F'); put(0); put(1); put(1); put(0); put16(1); put16(1); put(0); put(0);
    for (int t = 0; t < 2; t++) { put16(0xFFDB); put16(67); put(t); for (int k = 0; k < 64; k++) put(e->hq.q[t][zz[k]]); }
    put16(0xFFC2); put16(17); put(8); put16(g.H); put16(g.W); put(3);
    put(1); put((g.hs << 4) | g.vs); put(0); put(2); put(0x11); put(1); put(3); put(0x11); put(1);
  }
  const DeviceTables &ht = e->h_prog_tab[i];
  auto dht = [&](int w, int tcth) {
    const int nv = (int)ht.nvals[w];
    put16(0xFFC4); put16(19 + nv); put(tcth);
    for (int k = 1; k <= 16; k++) put(ht.bits[w][k]);
    for (int k = 0; k < nv; k++) put(ht.vals[w][k]);
  };
  if (q.sd.kind == 1) { dht(0, 0x00); dht(2, 0x01); }
  else if (q.sd.kind >= 3) dht(q.sd.comp[0] ? 3 : 1, 0x10 | (q.sd.comp[0] ? 1 : 0));
  if (i == 0) { put16(0xFFDD); put16(4); put16(g.ri); }
  put16(0xFFDA); put16(6 + 2 * q.sd.ncomp); put(q.sd.ncomp);
  for (int k = 0; k < q.sd.ncomp; k++) {
    const int c = q.sd.comp[k], t = c ? 1 : 0;
    put(c + 1);
    put(q.sd.Ss == 0 ? (q.Ah == 0 ? (t << 4) : 0) : t);   // jcmarker.c emit_sos: only the table kind the scan uses
  }
  put(q.sd.Ss); put(q.sd.Se); put((q.Ah << 4) | q.sd.Al);
}
// scan i's restart intervals, stuffed, back to back at `dst` (K6; RSTn behind every one of them, numbered from the scan's first
// interval in the WHOLE image: q.seg0 counts the intervals of the strips in front)
static int prog_compact(mij_encoder *e, int i, uint8_t *dst, size_t room) {
  const mij_encoder::ProgScan &q = e->ps[i];
  Geom g2 = e->g; g2.last_strip = 0; g2.mcu_first = 0; g2.seg0 = (uint32_t)q.seg0;       // RSTn numbering restarts in every scan, never an EOI
  HIPCHK(e, launch_compact(g2, q.scratch, q.slot, q.seg_bytes, q.seg_off, q.chunk_base, q.nseg, dst, room, q.res, e->prog_stream[q.stream_index]));
  return MIJ_OK;
}

static int encode_progressive(mij_encoder *e, hipStream_t s) {
  HIPCHK(e, hipEventRecord(e->prog_ev[4], s));            // the coefficients (K1 on `s`) are ready
  for (auto &st : e->prog_stream) HIPCHK(e, hipStreamWaitEvent(st, e->prog_ev[4], 0));
  int order[10], owner[10], rc;
  prog_plan(e, order, owner);
  for (int k = 0; k < 10; k++) {
    const int i = order[k];
    hipStream_t st = e->prog_stream[owner[i]];
    e->ps[i].stream_index = owner[i];
    if ((rc = prog_gather(e, i, st)) || (rc = prog_tables_emit(e, i, st))) return rc;
  }
  for (auto &st : e->prog_stream) HIPCHK(e, hipStreamSynchronize(st));
  prog_note_recoded(e);

  // ---- headers and offsets
  std::vector<uint8_t> hdr[10];
  size_t off = 0, data_off[10];
  for (int i = 0; i < 10; i++) {
    prog_header(e, i, hdr[i]);
    off += hdr[i].size();
    data_off[i] = off;
    const size_t sb = (size_t)e->h_prog_res[i].scan_bytes;
    if (sb < 2) return fail(e, MIJ_ERR_OVERFLOW, "progressive scan produced no data");
    off += sb - 2;                          // K6 ends every interval with RSTn; the last one of a scan has none: overwritten
  }
  if (off + 4 > e->capacity) {
    // larger than the preallocated buffer (noise at q100 with a restart marker after every block in every scan): all
    // sizes are known before anything is placed, so simply make room
    const size_t need = off + 65536;
    uint8_t *nb = nullptr;
    if (hipMalloc(&nb, HDR_AREA + need + 64) != hipSuccess) return fail(e, MIJ_ERR_OVERFLOW, "cannot grow the output buffer");
    (void)hipFree(e->d_out);
    e->d_out = nb; e->capacity = need;
  }
  uint8_t *const file = e->d_out + HDR_AREA;
  // compactions first (each one's trailing RSTn lands on the next scan's header position), then the headers on top
  for (int i = 0; i < 10; i++)
    if ((rc = prog_compact(e, i, file + data_off[i], e->capacity - data_off[i]))) return rc;
  for (int q4 = 0; q4 < 4; q4++) {
    HIPCHK(e, hipEventRecord(e->prog_ev[q4], e->prog_stream[q4]));
    HIPCHK(e, hipStreamWaitEvent(s, e->prog_ev[q4], 0));
  }
  for (int i = 0; i < 10; i++)
    HIPCHK(e, hipMemcpyAsync(file + data_off[i] - hdr[i].size(), hdr[i].data(), hdr[i].size(), hipMemcpyHostToDevice, s));
  static const uint8_t eoi[2] = {0xFF, 0xD9};
  HIPCHK(e, hipMemcpyAsync(file + off, eoi, 2, hipMemcpyHostToDevice, s));
  off += 2;
  if (e->timed_run) for (int i = 3; i <= 6; i++) HIPCHK(e, hipEventRecord(e->ev[i], s));   // stage [2] = all ten scans, [6] = total
  HIPCHK(e, hipStreamSynchronize(s));       // `hdr` lives on this stack frame
  e->h_res->scan_bytes = off; e->h_res->header_bytes = 0; e->h_res->flags = 0;
  e->issued = true;
  return MIJ_OK;
}

// ---- progressive output in strips (round 5; SURVEY.md 8e for the reference's own encoding, ImageCompressorImpl.cu:28) ------------
// With a restart interval that divides the MCU row every scan of libjpeg's script is strip-separable: a strip of MCU rows is a
// whole number of restart intervals in each of the ten scans, DC prediction and end-of-band runs stop at interval boundaries, and
// the only image-wide quantity is each scan's symbol statistics. So N ranks write the 1-rank file with ONE collective:
//   every rank:  mij_encode_transform; mij_encode_prog_statistics            (the ten scans' counts -> one contiguous device buffer)
//   caller:      all-reduce (sum) of mij_prog_histogram_buffer over the ranks
//   every rank:  mij_encode_prog_emit -> sizes[10], header_bytes[10]         (tables from the image-wide counts; this strip's intervals)
//   caller:      all-gather of sizes; file offset of (scan i, rank r) = sum of headers 0..i + sizes of scans < i + sizes[r' < r][i]
//   every rank:  mij_encode_prog_place(offsets, d_file, ...)                  (its ten segments to their places; flags say who writes
//                                                                             the headers -- the first strip's rank -- and the EOI)
// One rank with the whole image gives, through the same three calls, what mij_encode_entropy gives (the tests compare them).
int mij_encode_prog_statistics(mij_encoder *e, void *stream) {
  if (!e) return MIJ_ERR_INVALID_ARG;
  if (!e->p.progressive) return fail(e, MIJ_ERR_INVALID_ARG, "mij_encode_prog_statistics needs a progressive encoder");
  if (!e->transformed) return fail(e, MIJ_ERR_NOT_READY, "mij_encode_prog_statistics called before mij_encode_transform");
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  if (s != e->last_stream) HIPCHK(e, hipStreamWaitEvent(s, e->ev_xdone, 0));
  e->last_stream = s;
  int order[10], owner[10], rc;
  prog_plan(e, order, owner);
  // (DC refinement scans gather nothing: their slices stay zero, a sum over the ranks leaves them so)
  HIPCHK(e, hipMemsetAsync(e->d_prog_hist, 0, 10 * 4 * 257 * sizeof(uint32_t), s));
  HIPCHK(e, hipEventRecord(e->prog_ev[4], s));
  for (auto &st : e->prog_stream) HIPCHK(e, hipStreamWaitEvent(st, e->prog_ev[4], 0));
  for (int k = 0; k < 10; k++) {
    const int i = order[k];
    e->ps[i].stream_index = owner[i];
    if ((rc = prog_gather(e, i, e->prog_stream[owner[i]]))) return rc;
  }
  for (int q4 = 0; q4 < 4; q4++) {          // the caller's collective, on `s`, comes behind all of them
    HIPCHK(e, hipEventRecord(e->prog_ev[q4], e->prog_stream[q4]));
    HIPCHK(e, hipStreamWaitEvent(s, e->prog_ev[q4], 0));
  }
  e->prog_stats_done = true; e->prog_emitted = false;
  return MIJ_OK;
}

int mij_prog_histogram_buffer(mij_encoder *e, void **d_ptr, size_t *words) {
  if (!e || !d_ptr || !words) return MIJ_ERR_INVALID_ARG;
  if (!e->p.progressive || !e->d_prog_hist) return fail(e, MIJ_ERR_INVALID_ARG, "not a progressive encoder");
  *d_ptr = e->d_prog_hist; *words = 10 * 4 * 257;
  return MIJ_OK;
}

int mij_encode_prog_emit(mij_encoder *e, void *stream, uint64_t sizes[10], uint64_t header_bytes[10]) {
  if (!e || !sizes || !header_bytes) return fail(e, MIJ_ERR_INVALID_ARG, "null argument");
  if (!e->p.progressive || !e->prog_stats_done) return fail(e, MIJ_ERR_NOT_READY, "mij_encode_prog_emit follows mij_encode_prog_statistics");
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  e->last_stream = s;
  HIPCHK(e, hipEventRecord(e->prog_ev[4], s));            // behind the caller's all-reduce of the statistics
  for (auto &st : e->prog_stream) HIPCHK(e, hipStreamWaitEvent(st, e->prog_ev[4], 0));
  int rc;
  for (int i = 0; i < 10; i++)
    if ((rc = prog_tables_emit(e, i, e->prog_stream[e->ps[i].stream_index]))) return rc;
  for (auto &st : e->prog_stream) HIPCHK(e, hipStreamSynchronize(st));
  prog_note_recoded(e);
  std::vector<uint8_t> h;
  for (int i = 0; i < 10; i++) {
    prog_header(e, i, h);
    header_bytes[i] = h.size();
    const size_t sb = (size_t)e->h_prog_res[i].scan_bytes;
    if (e->ps[i].nseg > 0 && sb < 2) return fail(e, MIJ_ERR_OVERFLOW, "progressive scan produced no data");
    // every interval ends with its RSTn (K6); the file's last interval of a scan has none: the last strip's segment is 2 bytes shorter
    sizes[i] = e->g.last_strip ? sb - 2 : sb;
  }
  e->prog_stats_done = false; e->prog_emitted = true;
  return MIJ_OK;
}

int mij_encode_prog_place(mij_encoder *e, const uint64_t offsets[10], void *d_file, size_t file_capacity, uint64_t file_bytes, int flags,
                          void *stream) {
  if (!e || !offsets) return fail(e, MIJ_ERR_INVALID_ARG, "null argument");
  if (!e->p.progressive || !e->prog_emitted) return fail(e, MIJ_ERR_NOT_READY, "mij_encode_prog_place follows mij_encode_prog_emit");
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  uint8_t *file = (uint8_t *)d_file;
  size_t cap = file_capacity;
  if (!file) {           // this handle's own buffer (one rank, or a root that assembles at home): make room first, all sizes are known
    if (file_bytes + 4 > e->capacity) {
      const size_t need = (size_t)file_bytes + 65536;
      uint8_t *nb = nullptr;
      if (hipMalloc(&nb, HDR_AREA + need + 64) != hipSuccess) return fail(e, MIJ_ERR_OVERFLOW, "cannot grow the output buffer");
      (void)hipFree(e->d_out);
      e->d_out = nb; e->capacity = need;
    }
    file = e->d_out + HDR_AREA; cap = e->capacity;
  }
  std::vector<uint8_t> hdr[10];
  int rc;
  for (int i = 0; i < 10; i++) {
    const size_t sb = (size_t)e->h_prog_res[i].scan_bytes;
    if (offsets[i] + sb > cap) return fail(e, MIJ_ERR_OVERFLOW, "a scan's segment does not fit the target at its offset");
    if (flags & MIJ_PROG_PLACE_HEADERS) {
      prog_header(e, i, hdr[i]);
      if (hdr[i].size() > offsets[i]) return fail(e, MIJ_ERR_INVALID_ARG, "offset of a scan's first segment leaves no room for its header");
    }
  }
  // compactions first (each one's trailing RSTn may land on what follows: the next scan's header, the next rank's segment is NOT touched --
  // only the file's last interval of a scan loses its marker, and that one is followed by a header or the EOI), then the headers on top
  for (int i = 0; i < 10; i++)
    if (e->ps[i].nseg > 0 && (rc = prog_compact(e, i, file + offsets[i], cap - (size_t)offsets[i]))) return rc;
  for (int q4 = 0; q4 < 4; q4++) {
    HIPCHK(e, hipEventRecord(e->prog_ev[q4], e->prog_stream[q4]));
    HIPCHK(e, hipStreamWaitEvent(s, e->prog_ev[q4], 0));
  }
  if (flags & MIJ_PROG_PLACE_HEADERS)
    for (int i = 0; i < 10; i++)
      HIPCHK(e, hipMemcpyAsync(file + offsets[i] - hdr[i].size(), hdr[i].data(), hdr[i].size(), hipMemcpyHostToDevice, s));
  if (flags & MIJ_PROG_PLACE_EOI) {
    if (file_bytes < 2 || file_bytes > cap) return fail(e, MIJ_ERR_INVALID_ARG, "file_bytes does not fit the target");
    static const uint8_t eoi[2] = {0xFF, 0xD9};
    HIPCHK(e, hipMemcpyAsync(file + file_bytes - 2, eoi, 2, hipMemcpyHostToDevice, s));
  }
  HIPCHK(e, hipStreamSynchronize(s));       // `hdr` lives on this stack frame
  if (!d_file) { e->h_res->scan_bytes = file_bytes; e->h_res->header_bytes = 0; e->h_res->flags = 0; e->issued = true; }
  e->prog_emitted = false;
  return MIJ_OK;
}

int mij_encode_tables(mij_encoder *e, void *stream) {
  if (!e) return MIJ_ERR_INVALID_ARG;
  if (!e->transformed) return fail(e, MIJ_ERR_NOT_READY, "mij_encode_tables called before mij_encode_transform");
  if (e->p.progressive) return fail(e, MIJ_ERR_INVALID_ARG, "progressive output builds a table per scan inside mij_encode_entropy");
  if (e->tables_early) return MIJ_OK;
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(e, hipStreamWaitEvent(s, e->ev_xdone, 0));
  if (e->p.optimized_huffman || !e->static_tables_ready) {
    int rc = run_tail(e, s, true);
    if (rc) return rc;
    e->static_tables_ready = true;
  }
  if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[3], s));
  HIPCHK(e, hipEventRecord(e->ev_tab, s));
  e->tables_early = true;
  return MIJ_OK;
}

int mij_encode_entropy(mij_encoder *e, void *stream) {
  if (!e) return MIJ_ERR_INVALID_ARG;
  if (!e->transformed) return fail(e, MIJ_ERR_NOT_READY, "mij_encode_entropy called before mij_encode_transform");
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  if (s != e->last_stream) HIPCHK(e, hipStreamWaitEvent(s, e->ev_xdone, 0));     // entropy stage on another stream than the transform
  e->last_stream = s;
  const Geom &g = e->g;
  if (e->p.progressive) return encode_progressive(e, s);
  if (e->tables_early) {
    HIPCHK(e, hipStreamWaitEvent(s, e->ev_tab, 0));
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[7], s));      // the entropy stage starts here, not behind the tables
  } else {
    // Fixed (Annex K) tables and the header do not depend on the image: built once per handle.
    if (e->p.optimized_huffman || !e->static_tables_ready) {
      int rc = run_tail(e, s, true);
      if (rc) return rc;
      e->static_tables_ready = true;
    }
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[3], s));
  }
  e->fused_run = e->fuse;
  if (e->fuse) {
    // K4 with the size scan and the stuffing + compaction folded in (k_encode<1, true>): stage times [4], [5] read 0
    HIPCHK(e, launch_encode_fused(g, e->d_coef, e->d_tab, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_ff, e->nseg, e->d_status,
                                  e->d_redo, e->d_out + HDR_AREA, e->capacity, e->d_res, nullptr, s));
    if (e->timed_run) for (int i = 4; i <= 6; i++) HIPCHK(e, hipEventRecord(e->ev[i], s));
  } else {
    HIPCHK(e, launch_encode(g, e->d_coef, e->d_tab, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_ff, e->nseg, e->k4_narrow ? 2 : 0, s));
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[4], s));
    HIPCHK(e, launch_scan(e->d_seg_bytes, e->d_seg_ff, e->d_seg_off, e->nseg, e->d_chunk_total, e->d_chunk_base, e->d_ovf, e->d_res, s));
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[5], s));
    // (the compaction kernel also publishes the result record to the page-locked h_res: no device-to-host copy behind it)
    HIPCHK(e, launch_compact(g, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_off, e->d_chunk_base, e->nseg, e->d_out + HDR_AREA,
                             e->capacity, e->d_res, s, nullptr, e->h_res_dev));
    if (e->timed_run) HIPCHK(e, hipEventRecord(e->ev[6], s));
  }
  if (e->fuse || !e->h_res_dev) HIPCHK(e, hipMemcpyAsync(e->h_res, e->d_res, sizeof(DeviceResult), hipMemcpyDeviceToHost, s));
  HIPCHK(e, hipEventRecord(e->ev_done, s));
  e->issued = true;
  e->wait_event = true;
  e->collected = false;
  return MIJ_OK;
}

// ---- strip sharding without host round trips ------------------------------------
// The one-GPU entry points above learn the strip size on the host (mij_encode_result) -- fine for one GPU, a pipeline
// stall per image when N ranks must agree on offsets. These three keep sizes and offsets on the device:
//   mij_encode_entropy_sizes : tables + entropy coding + size scan; the strip's byte count lands in *d_size_slot
//   [caller all-gathers the slots of all ranks into d_sizes[world], on the device]
//   mij_encode_place         : compaction (K6) into this handle's own buffer; ranks > 0 then PUT the strip into rank 0's
//                              buffer (peer-mapped, see mij_ipc_*) at sum(d_sizes[0..rank)).
// The fast entropy coder leaves intervals with an oversized block to the roomy instantiation; here that one is always
// enqueued behind it (it exits at once for every other interval) so that no host decision sits in the pipeline.
int mij_encode_entropy_sizes(mij_encoder *e, uint64_t *d_size_slot, void *stream) {
  if (!e || !d_size_slot) return fail(e, MIJ_ERR_INVALID_ARG, "null argument");
  if (!e->transformed) return fail(e, MIJ_ERR_NOT_READY, "mij_encode_entropy_sizes called before mij_encode_transform");
  if (e->p.progressive) return fail(e, MIJ_ERR_INVALID_ARG, "progressive output is sharded through mij_encode_prog_statistics / _emit / _place");
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  e->last_stream = s;
  const Geom &g = e->g;
  if (e->p.optimized_huffman || !e->static_tables_ready) {
    int rc = run_tail(e, s, true);
    if (rc) return rc;
    e->static_tables_ready = true;
  }
  unsigned long long *slot = reinterpret_cast<unsigned long long *>(d_size_slot);
  if (e->fuse) {
    // fused coder; behind it the unfused kernels, GATED on its give-up flag: they fall through unless an interval held an
    // oversized block (rare), in which case they redo sizes and placement for the whole strip -- still without the host
    HIPCHK(e, launch_encode_fused(g, e->d_coef, e->d_tab, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_ff, e->nseg, e->d_status,
                                  e->d_redo, e->d_out + HDR_AREA, e->capacity, e->d_res, slot, s));
    HIPCHK(e, launch_encode(g, e->d_coef, e->d_tab, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_ff, e->nseg, 1, s, e->d_redo));
    HIPCHK(e, launch_scan(e->d_seg_bytes, e->d_seg_ff, e->d_seg_off, e->nseg, e->d_chunk_total, e->d_chunk_base, e->d_ovf, e->d_res, s, slot, e->d_redo));
  } else {
    HIPCHK(e, launch_encode(g, e->d_coef, e->d_tab, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_ff, e->nseg, e->k4_narrow ? 2 : 0, s));
    HIPCHK(e, launch_encode(g, e->d_coef, e->d_tab, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_ff, e->nseg, 1, s, nullptr, e->d_res));
    HIPCHK(e, launch_scan(e->d_seg_bytes, e->d_seg_ff, e->d_seg_off, e->nseg, e->d_chunk_total, e->d_chunk_base, e->d_ovf, e->d_res, s, slot));
  }
  e->timed_run = false;
  return MIJ_OK;
}

int mij_encode_place(mij_encoder *e, uint8_t *d_file_scan, size_t file_scan_capacity, const uint64_t *d_sizes, int rank, int world,
                     void *stream) {
  if (!e || !d_sizes || rank < 0 || rank >= world) return fail(e, MIJ_ERR_INVALID_ARG, "bad argument");
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  uint8_t *own_scan = e->d_out + HDR_AREA;
  // The rank whose buffer receives the file (the image's ROOT: d_file_scan null or its own scan area) compacts its strip
  // straight to its place in the file; every other rank compacts locally and puts the strip into the root's buffer.
  const bool is_root = !d_file_scan || d_file_scan == own_scan;
  const unsigned long long *szs = reinterpret_cast<const unsigned long long *>(d_sizes);
  e->place_timed = e->timing;
  if (e->place_timed && !e->ev_place_ok) {
    for (auto &v : e->ev_place) HIPCHK(e, hipEventCreate(&v));
    e->ev_place_ok = true;
  }
  if (e->place_timed) HIPCHK(e, hipEventRecord(e->ev_place[0], s));
  HIPCHK(e, launch_compact(e->g, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_off, e->d_chunk_base, e->nseg, own_scan, e->capacity,
                           e->d_res, s, e->fuse ? e->d_redo : nullptr, nullptr, is_root ? szs : nullptr, rank, world));      // fused: only if the fused placement was given up
  if (e->place_timed) HIPCHK(e, hipEventRecord(e->ev_place[1], s));
  if (!is_root)
    HIPCHK(e, launch_put(own_scan, szs, rank, world, d_file_scan, file_scan_capacity, e->capacity, e->d_res, s));
  if (e->place_timed) HIPCHK(e, hipEventRecord(e->ev_place[2], s));
  e->place_put = !is_root;
  e->placed_as_root = is_root;
  HIPCHK(e, hipEventRecord(e->ev_done, s));
  e->last_stream = s;
  e->issued = true;
  e->wait_event = true;
  e->sharded_pending = true;
  return MIJ_OK;
}

// Waits for the last mij_encode_place on this handle and reports the assembled file (rank 0: header + all strips, sizes
// read from d_sizes; other ranks: their own strip). MIJ_ERR_OVERFLOW if a strip or the file did not fit its buffer.
int mij_sharded_result(mij_encoder *e, const uint64_t *d_sizes, int rank, int world, mij_result *o) {
  if (!e || !d_sizes || !o || world < 1 || world > 4096) return fail(e, MIJ_ERR_INVALID_ARG, "bad argument");
  if (!e->sharded_pending) return fail(e, MIJ_ERR_NOT_READY, "no mij_encode_place has been issued on this handle");
  HIPCHK(e, hipSetDevice(e->p.device));
  HIPCHK(e, hipEventSynchronize(e->ev_done));
  std::vector<uint64_t> sz((size_t)world);
  HIPCHK(e, hipMemcpy(sz.data(), d_sizes, sz.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
  HIPCHK(e, hipMemcpy(e->h_res, e->d_res, sizeof(DeviceResult), hipMemcpyDeviceToHost));
  note_recoded(e);
  if (e->h_res->flags & 3u) return fail(e, MIJ_ERR_OVERFLOW, "a strip or the assembled file exceeds its output buffer (mij_encoder_reserve_output)");
  if (e->h_res->scan_bytes > e->capacity) return fail(e, MIJ_ERR_OVERFLOW, "strip exceeds this handle's output buffer");
  uint64_t total = 0;
  for (int r = 0; r < world; r++) total += sz[(size_t)r];
  if (e->placed_as_root && total > e->capacity) return fail(e, MIJ_ERR_OVERFLOW, "assembled file exceeds the root's output buffer (mij_encoder_reserve_output)");
  const size_t hb = e->h_res->header_bytes;
  o->d_buffer = e->d_out;
  o->header_offset = HDR_AREA - hb;
  o->header_bytes = hb;
  o->scan_offset = HDR_AREA;
  o->scan_bytes = e->placed_as_root ? (size_t)total : (size_t)sz[(size_t)rank];
  o->file_bytes = hb + o->scan_bytes;
  return MIJ_OK;
}

// Device times of the last mij_encode_place on this handle (timing enabled before it, mij_sharded_result called since):
// [0] stuffing + compaction (K6), [1] the put into the root's buffer (0 on the root itself, which has nothing to send).
int mij_place_times(mij_encoder *e, float ms[2]) {
  if (!e || !ms) return MIJ_ERR_INVALID_ARG;
  if (!e->place_timed || !e->ev_place_ok) return fail(e, MIJ_ERR_NOT_READY, "timing was not enabled for the last mij_encode_place");
  HIPCHK(e, hipSetDevice(e->p.device));
  HIPCHK(e, hipEventSynchronize(e->ev_place[2]));
  for (int i = 0; i < 2; i++) {
    float t = 0;
    if (hipEventElapsedTime(&t, e->ev_place[i], e->ev_place[i + 1]) != hipSuccess) t = -1.f;
    ms[i] = t;
  }
  if (!e->place_put) ms[1] = 0.f;
  return MIJ_OK;
}

// Room for `scan_capacity` bytes of entropy-coded data in this handle's output buffer (rank 0 of a sharded encode holds the
// whole file, not just its strip). Call before the first encode and before exporting the buffer.
int mij_encoder_reserve_output(mij_encoder *e, size_t scan_capacity) {
  if (!e) return MIJ_ERR_INVALID_ARG;
  if (scan_capacity <= e->capacity) return MIJ_OK;
  HIPCHK(e, hipSetDevice(e->p.device));
  if (e->issued) HIPCHK(e, hipStreamSynchronize(e->last_stream));
  // This buffer is what OTHER GPUs write into (k_put through a peer mapping) and what this GPU then reads (mij_retrieve_bitstream,
  // a decode of the assembled file, a framework's copy kernel). The peers' stores arrive at this device's memory behind its L2s,
  // which may still hold lines of the same addresses from the handle's earlier images (its own compaction wrote there, its own
  // k_put read from there). So the buffer is allocated UNCACHED for its owner (hipDeviceMallocUncached: MTYPE UC, no line of it
  // ever lives in an L2 or L1 of this device) -- what RCCL does with its own peer-written buffers -- and visibility of a peer's
  // bytes then rests on two things only, both in program order: k_put's system-scope release before it retires, and the
  // collective the peer enqueues behind it, which this rank's stream waits for. A cached allocation would
  // lean on the fabric probing this device's L2 on remote writes; no multi-GPU run has ever tested that here.
  // MIJ_SHARED_OUT=cached keeps plain hipMalloc (A/B experiments on one GPU); if the uncached allocation or its export fails the
  // plain one is taken as well.
  uint8_t *nb = nullptr;
  const size_t bytes = HDR_AREA + scan_capacity + 64;
  const char *mode = getenv("MIJ_SHARED_OUT");
  bool uncached = !(mode && strcmp(mode, "cached") == 0);
  if (uncached) {
    hipIpcMemHandle_t probe;
    if (hipExtMallocWithFlags((void **)&nb, bytes, hipDeviceMallocUncached) != hipSuccess) { (void)hipGetLastError(); nb = nullptr; uncached = false; }
    else if (hipIpcGetMemHandle(&probe, nb) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(nb); nb = nullptr; uncached = false; }
  }
  if (!nb && hipMalloc(&nb, bytes) != hipSuccess) { (void)hipGetLastError(); return fail(e, MIJ_ERR_ALLOC, "cannot reserve the output buffer"); }
  (void)hipFree(e->d_out);
  e->d_out = nb; e->capacity = scan_capacity;
  e->out_uncached = uncached;
  e->static_tables_ready = false;      // the header lives in this buffer
  return MIJ_OK;
}

int mij_output_is_uncached(const mij_encoder *e) { return e && e->out_uncached ? 1 : 0; }

int mij_output_buffer(mij_encoder *e, void **d_buffer, size_t *scan_offset, size_t *scan_capacity) {
  if (!e || !d_buffer) return MIJ_ERR_INVALID_ARG;
  *d_buffer = e->d_out;
  if (scan_offset) *scan_offset = HDR_AREA;
  if (scan_capacity) *scan_capacity = e->capacity;
  return MIJ_OK;
}

// Peer mapping of a device allocation (hipIpcGetMemHandle / hipIpcOpenMemHandle): how ranks > 0 reach rank 0's output buffer.
// `handle64` is MIJ_IPC_HANDLE_BYTES of opaque data to ship between processes (e.g. in a broadcast).
int mij_ipc_export(const void *d_ptr, void *handle64) {
  if (!d_ptr || !handle64) return MIJ_ERR_INVALID_ARG;
  static_assert(sizeof(hipIpcMemHandle_t) <= MIJ_IPC_HANDLE_BYTES, "handle size");
  hipIpcMemHandle_t h;
  hipError_t he = hipIpcGetMemHandle(&h, const_cast<void *>(d_ptr));
  if (he != hipSuccess) return fail(nullptr, MIJ_ERR_HIP, "hipIpcGetMemHandle", he);
  memset(handle64, 0, MIJ_IPC_HANDLE_BYTES);
  memcpy(handle64, &h, sizeof h);
  return MIJ_OK;
}
int mij_ipc_open(int device, const void *handle64, void **d_ptr) {
  if (!handle64 || !d_ptr) return MIJ_ERR_INVALID_ARG;
  *d_ptr = nullptr;
  HIPCHK(nullptr, hipSetDevice(device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle64, sizeof h);
  hipError_t he = hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess);
  if (he != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, MIJ_ERR_HIP, "hipIpcOpenMemHandle", he); }
  return MIJ_OK;
}
int mij_ipc_close(void *d_ptr) {
  if (!d_ptr) return MIJ_OK;
  return hipIpcCloseMemHandle(d_ptr) == hipSuccess ? MIJ_OK : MIJ_ERR_HIP;
}

int mij_encode_device(mij_encoder *e, const void *d_src, size_t pitch, size_t plane_stride, int fmt, void *stream) {
  int rc = mij_encode_transform(e, d_src, pitch, plane_stride, fmt, stream);
  if (rc) return rc;
  return mij_encode_entropy(e, stream);
}

int mij_encode_result(mij_encoder *e, mij_result *o) {
  if (!e || !o) return MIJ_ERR_INVALID_ARG;
  if (!e->issued) return fail(e, MIJ_ERR_NOT_READY, "no encode has been issued on this handle");
  HIPCHK(e, hipSetDevice(e->p.device));
  if (e->wait_event) HIPCHK(e, hipEventSynchronize(e->ev_done));   // this encode's work only (later work on the stream may still run)
  else HIPCHK(e, hipStreamSynchronize(e->last_stream));
  e->collected = true;
  if (e->h_res->flags & 1u) {
    // Some block needed more than a fast-path strip (768 bits): the fast encoder left those intervals marked; code them
    // with the roomy instantiation, then redo scan + compaction.
    hipStream_t s = e->last_stream;
    HIPCHK(e, launch_encode(e->g, e->d_coef, e->d_tab, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_ff, e->nseg, 1, s, nullptr, e->d_res));
    HIPCHK(e, launch_scan(e->d_seg_bytes, e->d_seg_ff, e->d_seg_off, e->nseg, e->d_chunk_total, e->d_chunk_base, e->d_ovf, e->d_res, s));
    HIPCHK(e, launch_compact(e->g, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_off, e->d_chunk_base, e->nseg, e->d_out + HDR_AREA,
                             e->capacity, e->d_res, s));
    HIPCHK(e, hipMemcpyAsync(e->h_res, e->d_res, sizeof(DeviceResult), hipMemcpyDeviceToHost, s));
    HIPCHK(e, hipStreamSynchronize(s));
  }
  if (!e->p.progressive) note_recoded(e);
  if (e->h_res->scan_bytes > e->capacity) {
    // Output larger than the preallocated buffer (very high quality on noise): grow it and redo header + compaction.
    const size_t need = (size_t)e->h_res->scan_bytes + 65536;
    uint8_t *nb = nullptr;
    if (hipMalloc(&nb, HDR_AREA + need + 64) != hipSuccess) return fail(e, MIJ_ERR_OVERFLOW, "cannot grow the output buffer");
    (void)hipFree(e->d_out);
    e->d_out = nb; e->capacity = need;
    hipStream_t s = e->last_stream;
    HIPCHK(e, launch_build_tables(e->g, e->d_hist, e->p.optimized_huffman ? 1 : 0, e->d_qt, e->d_tab, e->d_out, e->d_res, s));
    if (e->fused_run)    // the fused coder keeps no per-interval offsets: K5 produces them for K6
      HIPCHK(e, launch_scan(e->d_seg_bytes, e->d_seg_ff, e->d_seg_off, e->nseg, e->d_chunk_total, e->d_chunk_base, e->d_ovf, e->d_res, s));
    HIPCHK(e, launch_compact(e->g, e->d_scratch, e->slot_bytes, e->d_seg_bytes, e->d_seg_off, e->d_chunk_base, e->nseg, e->d_out + HDR_AREA,
                             e->capacity, e->d_res, s));
    HIPCHK(e, hipStreamSynchronize(s));
  }
  if (e->timed_run) {
    // ev: 0 start, 1 after transform, 2 after statistics, 3 after tables, 4 after encode, 5 after scan, 6 after compact
    // (tables built early on another stream: the entropy stage is timed from ev[7], and the total is the sum of the stages --
    // the image's kernels are interleaved with another image's then)
    float sum = 0;
    for (int i = 0; i < 6; i++) {
      float t = 0;
      if (hipEventElapsedTime(&t, e->ev[(i == 3 && e->tables_early) ? 7 : i], e->ev[i + 1]) != hipSuccess) t = -1.f;
      e->ms[i] = t;
      sum += t;
    }
    float t = 0;
    if (hipEventElapsedTime(&t, e->ev[0], e->ev[6]) != hipSuccess) t = -1.f;
    e->ms[6] = e->tables_early ? sum : t;
  }
  const size_t hb = e->h_res->header_bytes;
  o->d_buffer = e->d_out;
  o->header_offset = HDR_AREA - hb;
  o->header_bytes = hb;
  o->scan_offset = HDR_AREA;
  o->scan_bytes = (size_t)e->h_res->scan_bytes;
  o->file_bytes = hb + o->scan_bytes;
  return MIJ_OK;
}

int mij_retrieve_bitstream(mij_encoder *e, uint8_t *data, size_t *length) {
  if (!e || !length) return MIJ_ERR_INVALID_ARG;
  mij_result r;
  int rc = mij_encode_result(e, &r);
  if (rc) return rc;
  if (!data) { *length = r.file_bytes; return MIJ_OK; }
  const size_t n = std::min(*length, r.file_bytes);
  HIPCHK(e, hipMemcpy(data, r.d_buffer + r.header_offset, n, hipMemcpyDeviceToHost));
  *length = n;
  return MIJ_OK;
}

int mij_host_alloc(void **ptr, size_t bytes) {
  if (!ptr) return MIJ_ERR_INVALID_ARG;
  *ptr = nullptr;
  return hipHostMalloc(ptr, bytes, hipHostMallocDefault) == hipSuccess ? MIJ_OK : MIJ_ERR_ALLOC;
}

void mij_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

// Host-resident image in, host-resident JFIF out (the reference's compress(): cv::split + 3 blocking pageable copies,
// ImageCompressorImpl.cu:272-277, then nvjpegEncodeImage, then a D2H copy, .cu:285-287). Here the interleaved image is
// uploaded once, in ranges of MCU rows of about HOST_CHUNK_BYTES, and stage A of each range runs while the next range
// is on the wire; everything after stage A needs the complete statistics and runs once. With a page-locked source
// (mij_host_alloc, or memory the caller registered) the uploads are true async DMA; with pageable memory the HIP
// runtime stages each range itself and the overlap with stage A still holds.
constexpr size_t HOST_CHUNK_BYTES = 32u << 20;

int mij_encode_host(mij_encoder *e, const uint8_t *src, size_t pitch, size_t plane_stride, int fmt, const uint8_t **jpeg,
                    size_t *jpeg_bytes) {
  if (!jpeg || !jpeg_bytes) return fail(e, MIJ_ERR_INVALID_ARG, "null argument");
  int rc = check_input(e, src, pitch, fmt);
  if (rc) return rc;
  HIPCHK(e, hipSetDevice(e->p.device));
  const Geom &g = e->g;
  mij_geometry geo;
  mij_encoder_geometry(e, &geo);
  const bool interleaved = fmt == MIJ_INPUT_RGBI || fmt == MIJ_INPUT_BGRI;
  const size_t bytes = interleaved ? pitch * (size_t)geo.strip_rows : plane_stride * 2 + pitch * (size_t)geo.strip_rows;
  if (bytes > e->d_src_bytes) {
    (void)hipFree(e->d_src);
    e->d_src = nullptr; e->d_src_bytes = 0;
    HIPCHK(e, hipMalloc(&e->d_src, bytes));
    e->d_src_bytes = bytes;
  }
  if (!e->host_streams) {
    HIPCHK(e, hipStreamCreateWithFlags(&e->s_copy, hipStreamNonBlocking));
    HIPCHK(e, hipStreamCreateWithFlags(&e->s_work, hipStreamNonBlocking));
    for (auto &v : e->ev_chunk) HIPCHK(e, hipEventCreateWithFlags(&v, hipEventDisableTiming));
    e->host_streams = true;
  }
  const int mcu_h = 8 * g.vs, total_rows = strip_mcu_rows(g);
  const int per = (int)std::max<size_t>(1, HOST_CHUNK_BYTES / (pitch * (size_t)mcu_h * (interleaved ? 1 : 3)));
  int idx = 0;
  for (int r0 = 0; r0 < total_rows; r0 += per, idx++) {
    const int rows = std::min(per, total_rows - r0);
    const size_t y0 = (size_t)r0 * mcu_h;
    const size_t y1 = std::min((size_t)geo.strip_rows, (size_t)(r0 + rows) * mcu_h);   // the last MCU row may be partial
    if (y1 > y0) {
      for (int pl = 0; pl < (interleaved ? 1 : 3); pl++) {
        const size_t off = (size_t)pl * plane_stride + y0 * pitch;
        HIPCHK(e, hipMemcpyAsync(e->d_src + off, src + off, (y1 - y0) * pitch, hipMemcpyHostToDevice, e->s_copy));
      }
    }
    HIPCHK(e, hipEventRecord(e->ev_chunk[idx & 1], e->s_copy));
    HIPCHK(e, hipStreamWaitEvent(e->s_work, e->ev_chunk[idx & 1], 0));
    rc = transform_rows(e, e->d_src, pitch, plane_stride, fmt, e->s_work, r0, rows, r0 == 0, r0 + rows >= total_rows);
    if (rc) return rc;
  }
  rc = mij_encode_entropy(e, e->s_work);
  if (rc) return rc;
  mij_result r;
  rc = mij_encode_result(e, &r);
  if (rc) return rc;
  if (r.file_bytes > e->h_out_cap) {
    if (e->h_out) (void)hipHostFree(e->h_out);
    e->h_out = nullptr; e->h_out_cap = 0;
    const size_t cap = r.file_bytes + r.file_bytes / 8 + 4096;
    HIPCHK(e, hipHostMalloc(&e->h_out, cap, hipHostMallocDefault));
    e->h_out_cap = cap;
  }
  HIPCHK(e, hipMemcpy(e->h_out, r.d_buffer + r.header_offset, r.file_bytes, hipMemcpyDeviceToHost));
  *jpeg = e->h_out;
  *jpeg_bytes = r.file_bytes;
  return MIJ_OK;
}

// ---- secondary ("difference map") compression (reference README.md:8; definition SURVEY.md 8a A9) -------------------------
// D = dec(enc(I)) is a pure function of the quantised coefficients, and those are still in the handle's coefficient buffer
// after mij_encode_transform: the inverse transform reads them there (k_idct_enc: the tiled, transposed layout as it is) and
// the upsampling + colour kernel subtracts from the original on its way out. Round 2 Huffman-decoded the file it had just
// written (7.2 of the 10.8 ms of BASELINE config 5) and ran a separate subtraction pass over I and D.
static int ensure_sec(mij_encoder *e, size_t bytes) {
  if (bytes <= e->d_sec_bytes) return MIJ_OK;
  if (e->issued || e->transformed) HIPCHK(e, hipStreamSynchronize(e->last_stream));     // the old buffer may still be in use
  (void)hipFree(e->d_sec); e->d_sec = nullptr; e->d_sec_bytes = 0;
  HIPCHK(e, hipMalloc(&e->d_sec, bytes));
  e->d_sec_bytes = bytes;
  return MIJ_OK;
}
static size_t plane_bytes(const Geom &g) {       // Y padded to whole MCUs + Cb + Cr (k_idct's planes)
  return (size_t)g.mcux * g.hs * 8 * g.mcuy * g.vs * 8 + 2 * ((size_t)g.mcux * 8 * g.mcuy * 8);
}

static int gain_to_shift(int gain) { return gain <= 1 ? 0 : gain == 2 ? 1 : gain == 4 ? 2 : gain == 8 ? 3 : -1; }

static int encode_residual(mij_encoder *e, const void *d_src, size_t pitch, size_t plane_stride, int fmt, void *d_dst, size_t dst_pitch,
                           size_t dst_plane_stride, void *stream, int gain_shift);

int mij_encode_residual_device(mij_encoder *e, const void *d_src, size_t pitch, size_t plane_stride, int fmt, void *d_dst, size_t dst_pitch,
                               size_t dst_plane_stride, void *stream) {
  return encode_residual(e, d_src, pitch, plane_stride, fmt, d_dst, dst_pitch, dst_plane_stride, stream, 0);
}

int mij_encode_residual_gain_device(mij_encoder *e, const void *d_src, size_t pitch, size_t plane_stride, int fmt, void *d_dst, size_t dst_pitch,
                                    size_t dst_plane_stride, int gain, void *stream) {
  const int sh = gain_to_shift(gain);
  if (sh < 0) return fail(e, MIJ_ERR_INVALID_ARG, "gain must be 1, 2, 4 or 8");
  return encode_residual(e, d_src, pitch, plane_stride, fmt, d_dst, dst_pitch, dst_plane_stride, stream, sh);
}

static int encode_residual(mij_encoder *e, const void *d_src, size_t pitch, size_t plane_stride, int fmt, void *d_dst, size_t dst_pitch,
                           size_t dst_plane_stride, void *stream, int gain_shift) {
  if (!e || !d_dst) return fail(e, MIJ_ERR_INVALID_ARG, "null argument");
  const bool interleaved = fmt == MIJ_INPUT_RGBI || fmt == MIJ_INPUT_BGRI;
  if (!interleaved && fmt != MIJ_INPUT_RGB && fmt != MIJ_INPUT_BGR) return fail(e, MIJ_ERR_INVALID_ARG, "unknown pixel format");
  const Geom &g = e->g;
  if (g.mcu_first != 0 || !g.last_strip) return fail(e, MIJ_ERR_INVALID_ARG, "the difference map works on whole images (chroma upsampling crosses strip boundaries)");
  if (!e->transformed) return fail(e, MIJ_ERR_NOT_READY, "mij_encode_residual_device needs the coefficients of a mij_encode_transform / mij_encode_device on this handle");
  const size_t row = (size_t)g.W * (interleaved ? 3 : 1);
  if (dst_pitch < row || (d_src && pitch < row)) return fail(e, MIJ_ERR_INVALID_ARG, "pitch smaller than a pixel row");
  HIPCHK(e, hipSetDevice(e->p.device));
  hipStream_t s = (hipStream_t)stream;
  // the planes live at the START of d_sec (mij_secondary_encode_host keeps its residual image behind them)
  int rc = ensure_sec(e, plane_bytes(g));
  if (rc) return rc;
  if (s != e->last_stream) HIPCHK(e, hipStreamWaitEvent(s, e->ev_xdone, 0));      // the coefficients were written on another stream
  const size_t ysz = (size_t)g.mcux * g.hs * 8 * g.mcuy * g.vs * 8, csz = (size_t)g.mcux * 8 * g.mcuy * 8;
  uint8_t *py = e->d_sec, *pcb = py + ysz, *pcr = pcb + csz;
  Geom gr = g;
  gr.res_shift = gain_shift;      // only the difference-map stores read it
  if (idct_color_supported(g, fmt)) HIPCHK(e, launch_idct_color(gr, e->d_coef, nullptr, e->d_qt, DcFix(), (uint8_t *)d_dst, dst_pitch, fmt, s, (const uint8_t *)d_src, pitch));
  else {
    HIPCHK(e, launch_idct_enc(g, e->d_coef, e->d_qt, py, pcb, pcr, s));
    HIPCHK(e, launch_upsample_color(gr, py, pcb, pcr, (uint8_t *)d_dst, dst_pitch, dst_plane_stride, fmt, s, (const uint8_t *)d_src, pitch, plane_stride));
  }
  return MIJ_OK;
}

// Both layers end to end, host memory in and out. `sp` (may be null = same settings, gain 1): the second layer's own quality /
// sampling (coded by a second handle kept inside `e`) and the gain applied to the difference before it is coded.
int mij_secondary_encode_host_ex(mij_encoder *e, const mij_secondary_params *sp, const uint8_t *src, size_t pitch, size_t plane_stride, int fmt,
                                 uint8_t *primary, size_t *primary_bytes, uint8_t *secondary, size_t *secondary_bytes) {
  if (!e || !src || !primary || !primary_bytes || !secondary || !secondary_bytes) return fail(e, MIJ_ERR_INVALID_ARG, "null argument");
  const Geom &g = e->g;
  if (g.mcu_first != 0 || !g.last_strip) return fail(e, MIJ_ERR_INVALID_ARG, "secondary compression works on whole images");
  int q2 = e->p.quality, css2 = e->p.css, sh = 0;
  if (sp) {
    if (sp->struct_size != sizeof(mij_secondary_params)) return fail(e, MIJ_ERR_INVALID_ARG, "mij_secondary_params.struct_size matches no known layout");
    if (sp->quality2 != 0) q2 = sp->quality2;
    if (sp->css2 >= 0) css2 = sp->css2;
    sh = gain_to_shift(sp->gain);
    if (sh < 0) return fail(e, MIJ_ERR_INVALID_ARG, "mij_secondary_params.gain must be 0 (= 1), 1, 2, 4 or 8");
    if (q2 < 1 || q2 > 100) return fail(e, MIJ_ERR_INVALID_ARG, "mij_secondary_params.quality2 must be 0 (= the first layer's) or 1..100");
  }
  mij_encoder *e2 = e;       // the handle that codes the second layer
  if (q2 != e->p.quality || css2 != e->p.css) {
    if (e->sec_enc && (e->sec_quality != q2 || e->sec_css != css2)) { mij_encoder_destroy(e->sec_enc); e->sec_enc = nullptr; }
    if (!e->sec_enc) {
      mij_encoder_params p2 = e->p;
      p2.quality = q2; p2.css = css2;
      if (css2 != e->p.css) p2.restart_interval = MIJ_RESTART_AUTO;      // another MCU size: the interval that suits it
      p2.strip_mcu_row0 = p2.strip_mcu_rows = 0;
      int rc2 = mij_encoder_create(&p2, &e->sec_enc);
      if (rc2) {       // keep the reason mij_encoder_create recorded (out of memory: the second handle holds workspaces of its own, see mi_jpeg.h)
        const char *why = mij_last_error(nullptr);
        const std::string msg = std::string("cannot create the second layer's encoder (quality2 / css2): ") + (why ? why : "?");
        return fail(e, rc2, msg.c_str());
      }
      e->sec_quality = q2; e->sec_css = css2;
    }
    e2 = e->sec_enc;
  }
  const uint8_t *j1 = nullptr; size_t n1 = 0;
  int rc = mij_encode_host(e, src, pitch, plane_stride, fmt, &j1, &n1);      // J1; the image stays in e->d_src, its coefficients in e->d_coef
  if (rc) return rc;
  const size_t cap1 = *primary_bytes, cap2 = *secondary_bytes;
  *primary_bytes = n1;
  if (n1 > cap1) { *secondary_bytes = 0; return fail(e, MIJ_ERR_OVERFLOW, "primary buffer too small"); }
  memcpy(primary, j1, n1);
  const bool interleaved = fmt == MIJ_INPUT_RGBI || fmt == MIJ_INPUT_BGRI;
  const size_t bytes = interleaved ? pitch * (size_t)g.H : plane_stride * 2 + pitch * (size_t)g.H;
  const size_t planes = (plane_bytes(g) + 255) & ~(size_t)255;
  rc = ensure_sec(e, planes + bytes);
  if (rc) return rc;
  uint8_t *d_res = e->d_sec + planes;
  hipStream_t s = e->last_stream;
  // R = clip(((I - D) << sh) + 128) with D = dec(J1) reconstructed from the coefficients J1 was coded from (no decode of the file)
  rc = encode_residual(e, e->d_src, pitch, plane_stride, fmt, d_res, pitch, plane_stride, s, sh);
  if (rc) return rc;
  // the second handle codes on the SAME stream, behind the difference map: no host wait between the layers
  rc = mij_encode_device(e2, d_res, pitch, plane_stride, fmt, s);     // J2 = enc(R)
  if (rc) return e2 == e ? rc : fail(e, rc, mij_last_error(e2));
  size_t n2 = 0;
  rc = mij_retrieve_bitstream(e2, nullptr, &n2);
  if (rc) return e2 == e ? rc : fail(e, rc, mij_last_error(e2));
  *secondary_bytes = n2;
  if (n2 > cap2) return fail(e, MIJ_ERR_OVERFLOW, "secondary buffer too small");
  rc = mij_retrieve_bitstream(e2, secondary, &n2);
  return (rc && e2 != e) ? fail(e, rc, mij_last_error(e2)) : rc;
}

int mij_secondary_encode_host(mij_encoder *e, mij_decoder *dec, const uint8_t *src, size_t pitch, size_t plane_stride, int fmt,
                              uint8_t *primary, size_t *primary_bytes, uint8_t *secondary, size_t *secondary_bytes) {
  if (e && dec && mij_decoder_device(dec) != e->p.device) return fail(e, MIJ_ERR_INVALID_ARG, "encoder and decoder are on different devices");
  return mij_secondary_encode_host_ex(e, nullptr, src, pitch, plane_stride, fmt, primary, primary_bytes, secondary, secondary_bytes);
}

int mij_stage_times(mij_encoder *e, float ms[MIJ_NUM_STAGE_TIMES]) {
  if (!e || !ms) return MIJ_ERR_INVALID_ARG;
  if (!e->timed_run) return fail(e, MIJ_ERR_NOT_READY, "timing was not enabled for the last encode");
  memcpy(ms, e->ms, sizeof(e->ms));
  return MIJ_OK;
}

int mij_debug_coefficients(mij_encoder *e, int16_t *dst, size_t count) {
  if (!e || !dst) return MIJ_ERR_INVALID_ARG;
  HIPCHK(e, hipSetDevice(e->p.device));
  HIPCHK(e, hipStreamSynchronize(e->last_stream));
  // the device buffer is tiled and transposed (mij_internal.h); callers get [mcu][block in MCU][64 zig-zag]
  std::vector<uint8_t> raw(e->coef_alloc);
  HIPCHK(e, hipMemcpy(raw.data(), e->d_coef, e->coef_alloc, hipMemcpyDeviceToHost));
  const size_t n = std::min(count, e->coef_count);
  for (size_t i = 0; i < n; i++) {
    const size_t blk = i >> 6;
    const int k = (int)(i & 63);
    const size_t off = coef_block_offset(e->g, (long long)(blk / e->g.bpm), (int)(blk % e->g.bpm)) + (size_t)(k >> 3) * COEF_PIECE_STRIDE + (size_t)(k & 7) * 2;
    memcpy(&dst[i], &raw[off], 2);
  }
  return MIJ_OK;
}

int mij_debug_tables(mij_encoder *e, uint8_t *dst) {
  if (!e || !dst) return MIJ_ERR_INVALID_ARG;
  HIPCHK(e, hipSetDevice(e->p.device));
  HIPCHK(e, hipStreamSynchronize(e->last_stream));
  DeviceTables t;
  HIPCHK(e, hipMemcpy(&t, e->d_tab, sizeof(t), hipMemcpyDeviceToHost));
  for (int i = 0; i < 4; i++) { memcpy(dst + i * 273, t.bits[i], 17); memcpy(dst + i * 273 + 17, t.vals[i], 256); }
  return MIJ_OK;
}

int mij_copy_bench_device(void *d_dst, const void *d_src, size_t bytes, void *stream) {
  if (!d_dst || !d_src || (bytes & 15) || ((uintptr_t)d_dst & 15) || ((uintptr_t)d_src & 15)) return MIJ_ERR_INVALID_ARG;
  hipError_t he = launch_copy16(d_dst, d_src, bytes, (hipStream_t)stream);
  if (he != hipSuccess) return fail(nullptr, MIJ_ERR_HIP, "k_copy16 launch", he);
  return MIJ_OK;
}

int mij_clock_probe_device(int iters, void *stream, double *valu_clock_mhz, double *counter_clock_mhz, double *launch_ms) {
  if (iters <= 0 || iters > (1 << 20) || !valu_clock_mhz) return MIJ_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int wgs = clock_probe_workgroups();
  unsigned long long *d_out = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  std::vector<unsigned long long> h((size_t)wgs * 2 + 2);
  hipError_t he = hipMalloc((void **)&d_out, h.size() * sizeof(unsigned long long));
  if (he != hipSuccess) return fail(nullptr, MIJ_ERR_ALLOC, "clock probe buffer", he);
  float ms = 0.f;
  int dev = 0, wall_khz = 0;
  he = hipEventCreate(&e0);
  if (he == hipSuccess) he = hipEventCreate(&e1);
  if (he == hipSuccess) he = launch_clock_probe(d_out, (uint32_t *)(d_out + (size_t)wgs * 2), wgs, 16, s);      // warm-up: code object, clocks
  if (he == hipSuccess) he = hipEventRecord(e0, s);
  if (he == hipSuccess) he = launch_clock_probe(d_out, (uint32_t *)(d_out + (size_t)wgs * 2), wgs, iters, s);
  if (he == hipSuccess) he = hipEventRecord(e1, s);
  if (he == hipSuccess) he = hipEventSynchronize(e1);
  if (he == hipSuccess) he = hipEventElapsedTime(&ms, e0, e1);
  if (he == hipSuccess) he = hipMemcpy(h.data(), d_out, (size_t)wgs * 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (he == hipSuccess) he = hipGetDevice(&dev);
  if (he == hipSuccess && hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) wall_khz = 0;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(d_out);
  if (he != hipSuccess) return fail(nullptr, MIJ_ERR_HIP, "clock probe", he);
  // 8 waves per SIMD, each issuing iters x 64 four-cycle instructions: cycles a SIMD spent = 8 x iters x 64 x 4
  *valu_clock_mhz = ms > 0.f ? 8.0 * iters * 64.0 * 4.0 / (ms * 1e-3) / 1e6 : 0.0;
  if (counter_clock_mhz) {
    // median over the workgroups of (shader-clock counter ticks) / (constant-rate counter ticks) x the constant rate
    std::vector<double> r;
    for (int i = 0; i < wgs; i++)
      if (h[2 * i + 1]) r.push_back((double)h[2 * i] / (double)h[2 * i + 1]);
    std::sort(r.begin(), r.end());
    *counter_clock_mhz = (r.empty() || wall_khz <= 0) ? 0.0 : r[r.size() / 2] * wall_khz / 1e3;
  }
  if (launch_ms) *launch_ms = ms;
  return MIJ_OK;
}

int mij_synth_image_device(void *d_dst, int width, int y0, int rows, size_t pitch, int bgr, void *stream) {
  if (!d_dst || width <= 0 || rows <= 0 || pitch < (size_t)width * 3) return MIJ_ERR_INVALID_ARG;
  hipError_t he = launch_synth((uint8_t *)d_dst, width, y0, rows, pitch, bgr, (hipStream_t)stream);
  if (he != hipSuccess) return fail(nullptr, MIJ_ERR_HIP, "k_synth launch", he);
  return MIJ_OK;
}

}  // extern "C"

// mij_internal.h -- shared declarations between the HIP kernels (mij_kernels.hip) and the C-ABI host code
// (mij_api.hip). Not part of the public boundary (that is include/mi_jpeg.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <algorithm>
#include <utility>

namespace mij {

// ---- data layout in HBM (DESIGN.md "Data layout") -------------------------------------------------------------
// coefficients : int16, zig-zag order, in TILES of 256 luma blocks (= 256 / nl MCUs, K1's unit of work) holding the tile's
//                luma blocks (slot m * nl + s for block s of the tile's MCU m) and then its Cb and Cr blocks (slot 256 +
//                comp * MPT + m). Slots are stored in GROUPS of 64, transposed: group = 8 rows of 1 KiB, row j = the 16-byte
//                piece j (coefficients 8j .. 8j+7) of each of the 64 blocks. Both K1 and the entropy coders work lane-per-
//                block, so a wave stores / loads one piece of 64 blocks as one contiguous KiB (coef_block_offset below);
//                a block's coefficient k sits at offset + (k >> 3) * 1024 + (k & 7) * 2. (Round 1 kept blocks contiguous
//                in scan order: K1 then staged every block through LDS to get 64-byte runs, the coders through LDS again
//                to get from 8 lanes per block back to one.)
// statistics   : uint32 [4][257]  DC luma, AC luma, DC chroma, AC chroma (index 256 unused on device).
// tables       : DeviceTables (below): bits/vals for the DHT segments + the encoder LUT.
// scratch      : uint8 [nseg][slot_bytes]  un-stuffed entropy-coded bytes of each restart interval.
// seg_bytes / seg_ff / seg_off : uint32/uint32/uint64 [nseg]
// out          : uint8 [HDR_AREA + capacity]  header right-aligned in the first HDR_AREA bytes, then the scan data.

constexpr int HDR_AREA = 2048;          // >= largest possible baseline header (SOI..SOS with four full DHTs = 1737 B)
constexpr int LUT_DC_L = 0, LUT_AC_L = 16, LUT_DC_C = 273, LUT_AC_C = 289, LUT_SIZE = 546;
constexpr int LUT_AC_NONE = 256;        // extra AC entry that is always 0: "this coefficient codes nothing"
constexpr uint32_t SEG_OVERFLOW = 0xFFFFFFFFu;  // seg_bytes marker: interval left to the slow encoder instantiation
constexpr int MAX_BLOCK_WORDS = 53;     // ceil((16+11 + 63*(16+10)) / 32) + 1 : worst-case bits of one block
constexpr int MAX_BLOCK_BYTES = 212;

struct Quant {            // natural (row-major) coefficient order; [0] luma [1] chroma
  float recip[2][64];     // quantiser reciprocals r' (see quant_magic in k_common.inc)
  uint16_t q[2][64];
};

struct DeviceTables {
  uint8_t bits[4][17];    // [DC luma, AC luma, DC chroma, AC chroma]
  uint8_t vals[4][256];
  uint32_t nvals[4];
  uint32_t lut[LUT_SIZE]; // (code << 5) | length, indexed LUT_xx + symbol
  // K4's coding table in its final form (round 4), written once per image by K3: entry = ((codelen + nb) << 27) | (code << nb) for
  // symbol (run << 4) | nb at word run * 24 + nb * 2 + (chroma ? 1 : 0) (384 words, nb = 0 columns zero), then the DC entries at
  // 384 + nb * 2 + chroma (32 words), then EOB luma / chroma, ZRL luma / chroma (416..419). Every K4 workgroup used to rebuild
  // these 416 words from `lut` (seven scattered loads, a convert, a store, per restart interval: 40,625 times per image); now
  // it copies them with two 16-byte loads per lane.
  __attribute__((aligned(16))) uint32_t tok[448];
};
constexpr int TOK_DC = 384, TOK_EOB = 416, TOK_ZRL = 418;

struct DeviceResult {     // written by the scan kernel, copied to pinned host memory
  uint64_t scan_bytes;
  uint32_t header_bytes;
  uint32_t flags;
  uint32_t recoded;       // running count (never reset) of intervals the roomy entropy coder had to take over
  uint32_t pad;
};

struct Geom {
  int W, H;               // full image
  int hs, vs, bpm, nl;    // luma sampling, blocks per MCU, luma blocks per MCU
  int mcux, mcuy;         // whole image
  int wib0, hib0;         // luma width / height in real blocks
  int crows;              // chroma rows produced by real downsampling = ceil(H / vs)
  int ri;                 // restart interval in MCUs
  long long mcu_first, mcu_count;  // this strip
  int y_origin;           // image row of the strip's first source row
  int last_strip;
  int quality;
  // derived (geom_finish): 64-bit integer division costs ~300 instructions on this hardware, so the kernels divide by
  // multiplication (k_common.inc div_magic) or step incrementally
  uint32_t mcux_magic;    // floor(2^32 / mcux), saturated
  uint32_t seg0;          // index of the strip's first restart interval = mcu_first / ri
  int nl_sh, mpt_sh;      // log2(nl), log2(MCUs per tile = 256 / nl)
  int res_shift;          // difference-map kernels only (RES): log2 of the second layer's gain, R = clip(((I - D) << res_shift) + 128)
};
inline void geom_finish(Geom &g) {
  const unsigned long long m = 0x100000000ull / (unsigned)g.mcux;
  g.mcux_magic = (uint32_t)(m > 0xFFFFFFFFull ? 0xFFFFFFFFull : m);
  g.seg0 = (uint32_t)(g.mcu_first / g.ri);
  g.nl_sh = g.nl == 1 ? 0 : g.nl == 2 ? 1 : g.nl == 4 ? 2 : 3;
  g.mpt_sh = 8 - g.nl_sh;
}
// Coefficient buffer (see "data layout" above): bytes per tile, tiles of a strip, and the byte offset of piece 0 of block `s`
// (0 .. bpm-1, luma first) of the strip's MCU `mcu`; piece j is 1024 * j further on.
constexpr int COEF_PIECE_STRIDE = 1024, COEF_GROUP_BYTES = 8192;
__host__ __device__ inline size_t coef_tile_bytes(const Geom &g) { return (size_t)(256 + (2 << g.mpt_sh)) * 128; }
__host__ __device__ inline long long coef_tiles(const Geom &g, long long mcus) { return (mcus + (1 << g.mpt_sh) - 1) >> g.mpt_sh; }
__host__ __device__ inline size_t coef_block_offset(const Geom &g, long long mcu, int s) {
  const long long tile = mcu >> g.mpt_sh;
  const int m = (int)(mcu & ((1 << g.mpt_sh) - 1));
  const int slot = s < g.nl ? (m << g.nl_sh) + s : 256 + ((s - g.nl) << g.mpt_sh) + m;
  return (size_t)tile * coef_tile_bytes(g) + (size_t)(slot >> 6) * COEF_GROUP_BYTES + (size_t)(slot & 63) * 16;
}

struct TransformArgs {
  const uint8_t *src; size_t pitch, plane_stride;
  float fA[3], fC[3];     // colour matrix rows (Y, Cb, Cr) for the first / third stored channel, times 2^-16; G is fixed
  int16_t *coef;          // the STRIP's coefficient buffer (tiles count from the strip's first MCU)
  long long range_skip;   // MCUs between the strip's first MCU and g.mcu_first of this call (a strip may be transformed in several ranges)
  const float *recip_dev; // Quant::recip in device memory, read with scalar loads at the point of use (by value in the kernel
                          // arguments the 128 values were parked in SGPRs for the whole kernel and spilled through v_writelane;
                          // regrouped for one 64-byte load per column pair they were 1.3 % slower than these 8-byte loads)
  uint32_t *hist;         // non-null: optimised Huffman, take AC statistics (rows 1 and 3 of the 4 x 257 table)
  int16_t *dc;            // compact DC array [strip blocks] (written when hist != null, or when write_dc is set)
  int write_dc;           // progressive output: the DC scans read the compact array instead of whole blocks
  int fold_dc;            // hist != null only: the DC-difference statistics are taken here too (rows 0 and 2), from the tile's DC
                          // terms while they sit in LDS, and the compact array is not written unless write_dc asks for it. The
                          // caller sets it when every tile starts a restart interval (interval divides the tile's MCU count,
                          // first MCU on an interval boundary) and no block is a dummy: then no prediction crosses a tile
};

// launchers (mij_kernels.hip)
hipError_t launch_transform(const Geom &g, const TransformArgs &a, int interleaved, hipStream_t s);
hipError_t launch_dc_stats(const Geom &g, const int16_t *dc, uint32_t *hist, hipStream_t s);
// zero_hist (may be null): 4 x 257 words the kernel clears on its way out -- the statistics buffer of the handle's NEXT image
hipError_t launch_build_tables(const Geom &g, const uint32_t *hist, int optimize, const Quant *qt, DeviceTables *tab,
                               uint8_t *out, DeviceResult *res, hipStream_t s, uint32_t *zero_hist = nullptr);
// mode: 0 = fast coder, 24-word strips; 2 = fast coder, 16-word strips (5 waves per SIMD); 1 = roomy coder on the intervals
// a fast one gave up on (counts them in res->recoded when res is given)
hipError_t launch_encode(const Geom &g, const int16_t *coef, const DeviceTables *tab, uint8_t *scratch,
                         size_t slot_bytes, uint32_t *seg_bytes, uint32_t *seg_ff, long long nseg, int mode, hipStream_t s,
                         const uint32_t *gate = nullptr, DeviceResult *res = nullptr);
hipError_t launch_encode_fused(const Geom &g, const int16_t *coef, const DeviceTables *tab, uint8_t *scratch, size_t slot_bytes,
                               uint32_t *seg_bytes, uint32_t *seg_ff, long long nseg, unsigned long long *status, uint32_t *redo,
                               uint8_t *out_scan, size_t capacity, DeviceResult *res, unsigned long long *size_slot, hipStream_t s);
hipError_t launch_scan(const uint32_t *seg_bytes, const uint32_t *seg_ff, unsigned long long *seg_off, long long nseg,
                       unsigned long long *chunk_total, unsigned long long *chunk_base, uint32_t *ovf_flag, DeviceResult *res,
                       hipStream_t s, unsigned long long *size_slot = nullptr, const uint32_t *gate = nullptr);
hipError_t launch_put(const uint8_t *src, const unsigned long long *sizes, int rank, int world, uint8_t *file_scan, size_t file_capacity,
                      size_t max_bytes, DeviceResult *res, hipStream_t s);
hipError_t launch_compact(const Geom &g, const uint8_t *scratch, size_t slot_bytes, const uint32_t *seg_bytes,
                          const unsigned long long *seg_off, const unsigned long long *chunk_base, long long nseg,
                          uint8_t *out_scan, size_t capacity, const DeviceResult *res, hipStream_t s, const uint32_t *gate = nullptr,
                          DeviceResult *host_res = nullptr, const unsigned long long *root_sizes = nullptr, int root_rank = 0,
                          int root_world = 0);     // root_sizes: the strip goes sum(root_sizes[0 .. root_rank)) bytes into out_scan
hipError_t launch_copy16(void *dst, const void *src, size_t bytes, hipStream_t s);
hipError_t launch_synth(uint8_t *dst, int W, int y0, int rows, size_t pitch, int bgr, hipStream_t s);
int clock_probe_workgroups();
hipError_t launch_clock_probe(unsigned long long *out, uint32_t *sink, int workgroups, int iters, hipStream_t s);

// ---- decode path (k_decode.inc) ----
struct DecTables {                 // built on the host from the file's DHT / DQT / SOF / SOS segments
  uint16_t look[4][512];           // 9-bit look-ahead: (length << 8) | symbol, 0 = code longer than 9 bits
  int32_t maxcode[4][18];          // largest code of each length (-1: none)
  int32_t valoff[4][17];           // valptr[l] - mincode[l]
  uint8_t vals[4][256];
  uint16_t q[4][64];               // dequantisation, natural order
  int32_t tq[3], td[3], ta[3];
};
hipError_t launch_find_restarts(const uint8_t *scan, size_t n, unsigned long long *chunk_cnt, unsigned long long *chunk_base,
                                unsigned long long *seg_pos, long long nseg, uint32_t *flag, DeviceResult *res, hipStream_t s);
hipError_t launch_huff_decode(const Geom &g, const uint8_t *scan, size_t n, const unsigned long long *seg_pos, long long nseg,
                              const DecTables *tab, int16_t *coef, uint32_t *err_flag, hipStream_t s);
// One scan of a file that the generic scan decoder handles (k_decode_scans.inc).
struct ScanDesc {
  int kind;            // 0 sequential (DC + AC), 1 DC first, 2 DC refinement, 3 AC first, 4 AC refinement
  int ncomp;           // components in this scan; > 1 means MCU-interleaved
  int comp[3];         // 0 = Y, 1 = Cb, 2 = Cr
  int td[3], ta[3];    // table slots in DecTables (id * 2 for DC, id * 2 + 1 for AC)
  int Ss, Se, Al;
  int ri;              // MCUs (of this scan) per restart interval; 0 = no restart markers
  int bw, bh;          // single-component scan: that component's block grid (T.81 A.2.2: ceil(samples / 8))
  long long nmcu;      // MCUs of this scan
  int px_g;            // parallel progressive decoder only (k_decode_prog.inc): blocks between two anchors of this refinement scan
                       // (px_anchor_spacing; 0 = not set)
};

hipError_t launch_scan_decode(const Geom &g, const ScanDesc &sd, const uint8_t *scan, size_t n, const unsigned long long *seg_pos,
                              long long nseg, const DecTables *tab, int16_t *coef, uint32_t *err_flag, hipStream_t s);
// px_ws / px_flags (may be null): workspace (px_workspace_bytes) and four device words for the parallel progressive decoder
// (k_decode_prog.inc), which then goes first; the wave is its fallback, decided on the device.
hipError_t launch_scan_decode_wave(const Geom &g, const ScanDesc &sd, const uint8_t *scan, size_t n, unsigned long long *cnt,
                                   unsigned long long *base, unsigned long long *clean_len, uint8_t *clean, const DecTables *tab,
                                   int16_t *coef, uint32_t *scratch_flag, DeviceResult *scratch_res, uint32_t *err_flag, hipStream_t s,
                                   uint8_t *px_ws = nullptr, uint32_t *px_flags = nullptr, bool same_dc_tables = false);
constexpr int PX_FLAG_WORDS = 12;     // device words of one scan's parallel progressive decode (k_decode_prog.inc): [0] fell back, [1] pass changed, [2..8] diagnostics / gates
size_t px_workspace_bytes(const ScanDesc &sd, const Geom &g, size_t raw_len);
bool px_supported(const ScanDesc &sd);
hipError_t launch_px_scan(const Geom &g, const ScanDesc &sd, const uint8_t *clean, const unsigned long long *clean_len, size_t raw_len,
                          const DecTables *tab, int16_t *coef, uint8_t *ws, uint32_t *flags, bool same_dc_tables, hipStream_t s);
// Progressive scans (k_encode_prog.inc): gather != 0 counts symbols into hist (4 x 257), otherwise writes the interval slots.
hipError_t launch_prog_encode(const Geom &g, const ScanDesc &sd, int gather, const int16_t *coef, const DeviceTables *tab, uint8_t *scratch,
                              size_t slot_bytes, uint32_t *seg_bytes, uint32_t *seg_ff, uint32_t *hist, long long nseg, hipStream_t s,
                              const uint8_t *only_flagged = nullptr);
// Lane-per-block form (k_encode_prog2.inc); seg_flag[interval] = 1 where the emit pass left an interval to the serial kernel.
// narrow (refinement scans, emit pass): 16-word strips, five waves per SIMD; overflowed intervals are counted in res->recoded.
bool prog2_supported(const ScanDesc &sd);
hipError_t launch_prog2(const Geom &g, const ScanDesc &sd, int gather, const int16_t *coef, const int16_t *dc, const DeviceTables *tab,
                        uint8_t *scratch, size_t slot_bytes, uint32_t *seg_bytes, uint32_t *seg_ff, uint32_t *hist, uint8_t *seg_flag,
                        long long nseg, hipStream_t s, bool narrow, DeviceResult *res);
// Parallel (self-synchronising) decode of a baseline interleaved scan. `ws` is a workspace of par_workspace_bytes().
size_t par_workspace_bytes(size_t scan_len, long long nseg);
// The decoder reads an un-stuffed copy of the scan without its restart markers: launch_clean_scan makes it (clean: capacity >=
// n + 64; cnt / base: 2 x (n / 16384 + 1) words each) together with the intervals' start positions in it.
hipError_t launch_clean_scan(const uint8_t *scan, size_t n, unsigned long long *cnt, unsigned long long *base, unsigned long long *seg_pos,
                             long long nseg, uint8_t *clean, unsigned long long *clean_len, uint32_t *flag, DeviceResult *res, hipStream_t s);
// The parallel decoder leaves every DC term as the sum of the differences since the start of ITS SUBSEQUENCE; what is missing
// -- the sum over the earlier subsequences of the restart interval -- is in the scanned per-subsequence sums, which `fix`
// describes (they live in `ws`): launch_idct adds it while it loads the block (DcFix::pref == nullptr: DC terms are final).
struct DcFix {
  const int4 *pref = nullptr;        // exclusive scan over the subsequences of (blocks begun, DC-difference sums Y, Cb, Cr)
  const int4 *sub_pref = nullptr;    // exclusive scan over the restart intervals of their subsequence counts (.x)
  uint32_t bpi = 0, bpi_magic = 0;   // blocks per restart interval, floor(2^32 / bpi)
};
hipError_t launch_par_decode(const Geom &g, const uint8_t *clean, size_t n, const unsigned long long *clean_len, const unsigned long long *seg_pos,
                             long long nseg, const DecTables *tab, int16_t *coef, void *ws, uint32_t *changed, uint32_t *err_flag, int *passes,
                             hipStream_t s, DcFix *fix);
hipError_t launch_idct(const Geom &g, const int16_t *coef, const DecTables *tab, uint8_t *py, uint8_t *pcb, uint8_t *pcr, hipStream_t s,
                       const DcFix &fix = DcFix{});
// orig != null: stores the difference map R = clip(orig - D + 128) instead of the decoded pixels D (orig laid out like dst)
hipError_t launch_upsample_color(const Geom &g, const uint8_t *py, const uint8_t *pcb, const uint8_t *pcr, uint8_t *dst, size_t pitch,
                                 size_t plane_stride, int out_fmt, hipStream_t s, const uint8_t *orig = nullptr, size_t orig_pitch = 0,
                                 size_t orig_plane_stride = 0);
// The inverse transform straight from an ENCODER's coefficient buffer (tiled, transposed) and quantisation table.
hipError_t launch_idct_enc(const Geom &g, const int16_t *coef, const Quant *qt, uint8_t *py, uint8_t *pcb, uint8_t *pcr, hipStream_t s);
// k_idct + k_upsample_color in one kernel where idct_color_supported() (k_decode.inc: k_idct_color)
bool idct_color_supported(const Geom &g, int out_fmt);
hipError_t launch_idct_color(const Geom &g, const int16_t *coef, const DecTables *tab, const Quant *qt, const DcFix &fix, uint8_t *dst, size_t pitch,
                             int out_fmt, hipStream_t s, const uint8_t *orig = nullptr, size_t orig_pitch = 0);
hipError_t launch_residual(const uint8_t *a, const uint8_t *b, uint8_t *out, size_t n, int sign, hipStream_t s, int gain_shift = 0);

}  // namespace mij

// mij_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4): the JPEG encode hot path.
//
// Replaces what the reference gets from nvjpegEncodeImage (reference ImageCompressorImpl.cu:280):
//   K1 k_transform     BGR/RGB -> YCbCr, chroma downsample, level shift, 8x8 FDCT, quantise, zig-zag   (SURVEY 8a A3+A4)
//   K2 k_dc_stats      DC-difference statistics when K1 cannot take them itself (the AC ones are always K1's)   (A5+A6)
//   K3 k_build_tables  optimal (or Annex K) Huffman tables, encoder LUT, JFIF header                   (A6+A7)
//   K4 k_encode        Huffman coding + bit packing, one restart interval per wavefront                (A5+A7)
//   K5 k_scan          exclusive scan of interval sizes                                                (A7)
//   K6 k_compact       FF00 byte stuffing + compaction + RSTn / EOI markers                            (A7)
//   KP2 k_prog2_dc/_ac  progressive (SOF2) scans: libjpeg's script, gather / emit per scan, LANE PER BLOCK, end-of-band runs resolved with wave ballots (k_encode_prog2.inc)
//   KP k_prog_encode   the same scans lane per restart interval: exact serial fallback for intervals KP2 hands back   (k_encode_prog.inc)
// and, for the decode half (nvjpegDecodeJpeg*, ImageCompressorImpl.cu:361-366; getCVImageOnCPU, .cu:184-232):
//   D1 k_rst_count/write   restart-marker positions                                                    (k_decode.inc)
//   D2p k_par_decode<0..3> subsequence-parallel, self-synchronising Huffman decode of baseline scans   (k_decode_par.inc)
//   D2 k_huff_decode       lane per restart interval (fallback)                                        (k_decode.inc)
//   D2s k_scan_decode      progressive / multi-scan / greyscale scans, lane per restart interval       (k_decode_scans.inc)
//   D2w k_scan_decode_wave the same scans when they have NO restart markers: one wave walks the chain   (k_decode_wave.inc)
//   D2x k_px_*             progressive scans without restart markers IN PARALLEL: first scans by subsequence synchronisation,
//                          refinement scans by hypothesis search + exact verification; D2w is its fallback (k_decode_prog.inc)
//   D3 k_idct, D4 k_upsample_color[8], k_residual   IDCT, upsampling + colour, difference map          (k_decode.inc)
// Bit-identity with stock JPEG codecs means following their integer procedures: the "islow" FDCT / IDCT factorisation and
// constants, the colour / downsampling rounding rules, the quantiser and the optimal-table and progressive procedures are
// those of the Independent JPEG Group's libjpeg and of libjpeg-turbo, re-implemented here for wave-64 hardware.
// This software is based in part on the work of the Independent JPEG Group.
// No MFMA anywhere: this path is integer/byte work bounded by HBM and by VALU issue, not a contraction.
// Wavefront = 64 lanes throughout.
#include "mij_internal.h"

namespace mij {

#include "k_common.inc"
#include "k_transform.inc"
#include "k_batch.inc"
#include "k_stats.inc"
#include "k_tables.inc"
#include "k_encode.inc"
#include "k_finish.inc"
#include "k_encode_prog.inc"
#include "k_encode_prog2.inc"
#include "k_synth.inc"
#include "k_decode.inc"
#include "k_decode_scans.inc"
#include "k_decode_wave.inc"
#include "k_decode_par.inc"
#include "k_launch.inc"
#include "k_decode_prog.inc"

}  // namespace mij

// mij_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X, CDNA4): the JPEG encode hot path.
//
// Replaces what the reference gets from nvjpegEncodeImage (reference ImageCompressorImpl.cu:280):
//   K1 k_transform     BGR/RGB -> YCbCr, chroma downsample, level shift, 8x8 FDCT, quantise, zig-zag   (SURVEY 8a A3+A4)
//   K2 k_histogram     DC-difference / AC run-length symbol statistics                                 (A5+A6)
//   K3 k_build_tables  optimal (or Annex K) Huffman tables, encoder LUT, JFIF header                   (A6+A7)
//   K4 k_encode        Huffman coding + bit packing, one restart interval per wavefront                (A5+A7)
//   K5 k_scan          exclusive scan of interval sizes                                                (A7)
//   K6 k_compact       FF00 byte stuffing + compaction + RSTn / EOI markers                            (A7)
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
#include "k_synth.inc"
#include "k_decode.inc"
#include "k_decode_scans.inc"
#include "k_decode_par.inc"
#include "k_launch.inc"

}  // namespace mij

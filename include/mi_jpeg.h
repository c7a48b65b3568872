/*
 * mi_jpeg.h -- C ABI of the MI355X-native JPEG path (libmijpeg.so).
 *
 * This is the drop-in boundary for the one hot path of OroChippw/Nvjpeg-ImageCompressor: everything the reference
 * does through NVIDIA nvJPEG between `NvjpegCompressRunnerImpl::initCompressEnv` and
 * `nvjpegEncodeRetrieveBitstream`. Each entry point names the reference interface it replaces
 * (paths relative to the reference root, src/ImageCompressorDll/).
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative MIJ_ERR_* code and
 * NEVER calls exit() (the reference's CHECK_CUDA / CHECK_NVJPEG do, ImageCompressorImpl.cuh:16-34);
 * `mij_last_error` returns a human-readable message for the last failure on that handle.
 * One handle per thread; a handle is not re-entrant (same contract as the reference: one encoder state and one set
 * of device planes per instance, ImageCompressorImpl.cuh:58-63).
 * "device pointer" = HIP device memory on the handle's device; `stream` = a hipStream_t passed as void* (NULL = the
 * null stream, which is what the reference encodes on, ImageCompressorImpl.cu:280).
 */
#ifndef MI_JPEG_H_
#define MI_JPEG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIJ_API __attribute__((visibility("default")))

/* Error codes */
enum {
  MIJ_OK = 0,
  MIJ_ERR_INVALID_ARG = -1,
  MIJ_ERR_HIP = -2,          /* a HIP runtime call failed (replaces CHECK_CUDA's exit(1)) */
  MIJ_ERR_NO_DEVICE = -3,    /* no usable gfx950 device: the library has NO CPU fallback */
  MIJ_ERR_NOT_READY = -4,    /* result requested before an encode was issued */
  MIJ_ERR_OVERFLOW = -5,     /* output exceeded the capacity that could be allocated */
  MIJ_ERR_BAD_STREAM = -6,   /* decoder: not a JPEG this decoder handles */
  MIJ_ERR_ALLOC = -7
};

/* Chroma subsampling: same integer values as nvjpegChromaSubsampling_t, so code that passes
 * NVJPEG_CSS_* ints keeps working (reference hard-codes NVJPEG_CSS_444, ImageCompressorImpl.cu:31). */
enum { MIJ_CSS_444 = 0, MIJ_CSS_422 = 1, MIJ_CSS_420 = 2, MIJ_CSS_440 = 3, MIJ_CSS_411 = 4, MIJ_CSS_410 = 5 };

/* Input layouts: same integer values as nvjpegInputFormat_t. Planar = three planes of `pitch` x height bytes,
 * plane p at src + p * plane_stride (the reference's nvjpegImage_t with pitch = width and NVJPEG_INPUT_BGR,
 * ImageCompressorImpl.cuh:61-62, .cu:33-37). Interleaved = cv::Mat CV_8UC3 rows (main.cpp:37). */
enum { MIJ_INPUT_RGB = 3, MIJ_INPUT_BGR = 4, MIJ_INPUT_RGBI = 5, MIJ_INPUT_BGRI = 6 };

#define MIJ_RESTART_AUTO (-1)

typedef struct mij_encoder mij_encoder;

/* ABI version of this header. mij_encoder_params starts with its own size so that a caller compiled against an older
 * (shorter) layout is recognised instead of being read past its end: fields the caller's struct does not have are taken
 * as 0; a size this library does not know is rejected (MIJ_ERR_INVALID_ARG). Use MIJ_ENCODER_PARAMS_INIT or set
 * struct_size = sizeof(mij_encoder_params). */
#define MIJ_ABI_VERSION 2

typedef struct mij_encoder_params {
  uint32_t struct_size;    /* = sizeof(mij_encoder_params) of the header the CALLER was compiled with */
  int width, height;       /* full image size; reference ctor args (ImageCompressor.h:27), default 8320 x 40000 */
  int quality;             /* 1..100, IJG scaling; nvjpegEncoderParamsSetQuality (ImageCompressorImpl.cu:30) */
  int optimized_huffman;   /* nvjpegEncoderParamsSetOptimizedHuffman (ImageCompressorImpl.cu:29) */
  int css;                 /* MIJ_CSS_*; nvjpegEncoderParamsSetSamplingFactors (ImageCompressorImpl.cu:31) */
  int restart_interval;    /* MCUs per restart interval (DRI); MIJ_RESTART_AUTO lets the library choose (an interval
                              whose block count is a whole number of 64-block batches: 64 MCUs at 4:2:2). The unit of
                              GPU parallelism is one restart interval, so 0 (= no restart markers) is rejected. */
  int device;              /* HIP device ordinal */
  /* Strip sharding (multi-GPU). This encoder handles MCU rows [strip_mcu_row0, +strip_mcu_rows) of the
   * full image; strip_mcu_rows == 0 means the whole image. The strip's first MCU must fall on a restart-interval
   * boundary: strip_mcu_row0 * mcus_per_row must be a multiple of the interval (any row when the interval divides
   * the MCUs per row; otherwise every lcm(interval, mcus_per_row) / mcus_per_row rows -- 8 rows at the headline size). */
  int strip_mcu_row0, strip_mcu_rows;
  /* nvjpegEncoderParamsSetEncoding (ImageCompressorImpl.cu:28): 0 = baseline sequential (SOF0, one scan), the default
   * and the fast path; 1 = progressive (SOF2): the same coefficients coded as the ten scans of libjpeg's default script
   * with an optimal Huffman table per scan -- byte-identical to libjpeg-turbo's progressive output, a few per cent
   * smaller than baseline, about three times slower to produce (a statistics and an emit pass per scan; see DESIGN.md section 5
   * for the measured times). Whole images, or strips when restart_interval divides the MCUs per row and the width is a whole
   * number of MCUs ("Progressive output in STRIPS" below); optimized_huffman is implied. */
  int progressive;
} mij_encoder_params;
#define MIJ_ENCODER_PARAMS_INIT {(uint32_t)sizeof(mij_encoder_params), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}

/* Geometry derived from the parameters (useful to callers that shard). */
typedef struct mij_geometry {
  int hs, vs;                  /* luma sampling factors */
  int mcu_w, mcu_h;            /* MCU size in pixels */
  int mcus_per_row, mcu_rows;  /* whole image */
  int blocks_per_mcu;
  int restart_interval;        /* resolved value */
  int64_t strip_first_mcu, strip_mcus;
  int strip_y0, strip_rows;    /* pixel rows of the full image this strip reads */
} mij_geometry;

/* Result of one encode on a strip (device resident; valid until the next encode on the handle). */
typedef struct mij_result {
  const uint8_t *d_buffer;   /* device buffer holding [header][entropy-coded data][EOI] */
  size_t header_offset;      /* where SOI sits inside d_buffer */
  size_t header_bytes;       /* SOI .. SOS header (same on every strip) */
  size_t scan_offset;        /* = header_offset + header_bytes */
  size_t scan_bytes;         /* entropy-coded bytes of this strip incl. RSTn markers between and after its intervals;
                                the last strip of the image ends with EOI instead of a trailing RSTn */
  size_t file_bytes;         /* header_bytes + scan_bytes: a complete JFIF file when the strip is the whole image */
} mij_result;

MIJ_API const char *mij_version(void);
MIJ_API int mij_abi_version(void);   /* MIJ_ABI_VERSION the library was built with */
/* SHA-256 (hex) over the sources, headers and compile flags this binary was built from (nvjpeg_imagecompressor_amd/build.py
 * source_hash()): lets a deployment check that the .so it ships matches the tree. */
MIJ_API const char *mij_source_hash(void);
MIJ_API int mij_device_count(void);

/* initCompressEnv (ImageCompressorImpl.cu:19-45): create handles, push parameters, allocate device workspace. */
MIJ_API int mij_encoder_create(const mij_encoder_params *params, mij_encoder **out);
/* destoryCompressEnv (ImageCompressorImpl.cu:47-65). NULL is a no-op. */
MIJ_API void mij_encoder_destroy(mij_encoder *enc);
MIJ_API int mij_encoder_geometry(const mij_encoder *enc, mij_geometry *out);
/* The same geometry from the parameters alone: pure arithmetic, no device, nothing allocated (what a sharding host needs
 * to cut strips before it creates its handle). */
MIJ_API int mij_geometry_query(const mij_encoder_params *params, mij_geometry *out);
MIJ_API const char *mij_last_error(const mij_encoder *enc);

/* nvjpegEncodeImage (ImageCompressorImpl.cu:280): device-resident pixels -> device-resident bitstream, asynchronous
 * on `stream`. d_src points at the first pixel row of THIS strip (row strip_y0 of the image). */
MIJ_API int mij_encode_device(mij_encoder *enc, const void *d_src, size_t pitch, size_t plane_stride, int input_format,
                              void *stream);

/* The same work split at its only cross-GPU exchange point (optimised Huffman statistics):
 *   mij_encode_transform : colour convert + downsample + FDCT + quantise (+ symbol statistics when optimised)
 *   [caller all-reduces the statistics returned by mij_histogram_device across ranks, in place]
 *   mij_encode_entropy   : Huffman table build + entropy coding + restart markers + headers */
MIJ_API int mij_encode_transform(mij_encoder *enc, const void *d_src, size_t pitch, size_t plane_stride,
                                 int input_format, void *stream);
MIJ_API int mij_encode_entropy(mij_encoder *enc, void *stream);
/* Optional, between the two: build this image's Huffman tables and header on `stream` -- which may be another stream than
 * the transform's; the call orders itself behind the transform and mij_encode_entropy orders itself behind it. The table
 * build is one workgroup for ~45 us: a caller with several images in flight (two handles) issues
 *   transform(A, main) ; tables(A, side) ; entropy(B, main) ; ...
 * so that it runs under B's entropy coder instead of leaving the device idle. Without this call mij_encode_entropy builds
 * the tables itself. No counterpart in the reference (nvjpegEncodeImage is one call, ImageCompressorImpl.cu:280). */
MIJ_API int mij_encode_tables(mij_encoder *enc, void *stream);
/* Device pointer to the 4 x 257 uint32 symbol statistics (DC luma, AC luma, DC chroma, AC chroma). */
MIJ_API int mij_histogram_device(mij_encoder *enc, uint32_t **d_hist, size_t *count);
/* Use caller-owned device memory (e.g. a framework tensor that a collective can reduce) for the statistics. */
MIJ_API int mij_set_histogram_buffer(mij_encoder *enc, uint32_t *d_hist);

/* ---- Strip sharding with sizes and offsets kept on the device (multi-GPU; the reference has no counterpart:
 * one nvjpegEncodeImage call on one GPU, ImageCompressorImpl.cu:280). Per image and rank:
 *   mij_encode_transform -> [all-reduce of the statistics] -> mij_encode_entropy_sizes(&d_slot)
 *   -> [all-gather of the slots into d_sizes[world]] -> mij_encode_place(...)
 * Nothing in that sequence waits on the host. One rank -- the image's ROOT, any rank that owns a strip -- assembles the file
 * in its own output buffer (reserve room for the whole file with mij_encoder_reserve_output) and passes d_file_scan = NULL
 * (or its own scan area) to mij_encode_place: its strip is compacted straight to byte offset sum(d_sizes[0..rank)) of the scan
 * area. Every other rank passes the root's scan area as it sees it through a peer mapping (mij_ipc_export on the root,
 * mij_ipc_open on the others) and writes its strip there at its own offset. The root may differ from image to image: a
 * root receives (world - 1) / world of every file it assembles over its inbound links, so rotating it spreads that load
 * (DESIGN.md section 7). The file is complete on the root once every rank's mij_encode_place has executed -- e.g. when a
 * later collective on the same streams completes. mij_sharded_result waits for this handle's part and reports the file
 * (the root of the handle's last image) / the strip (others). */
MIJ_API int mij_encode_entropy_sizes(mij_encoder *enc, uint64_t *d_size_slot, void *stream);
MIJ_API int mij_encode_place(mij_encoder *enc, uint8_t *d_file_scan, size_t file_scan_capacity, const uint64_t *d_sizes, int rank,
                             int world, void *stream);
MIJ_API int mij_sharded_result(mij_encoder *enc, const uint64_t *d_sizes, int rank, int world, mij_result *out);
/* Device times in ms of the last mij_encode_place (mij_encoder_enable_timing before it): [0] stuffing + compaction,
 * [1] the put of the strip into the root's peer-mapped buffer (0 on the root). Strip bytes / [1] = what one xGMI link gave. */
MIJ_API int mij_place_times(mij_encoder *enc, float ms[2]);
/* mij_encoder_reserve_output allocates the buffer device-UNCACHED (hipDeviceMallocUncached) when it can: other GPUs write into it
 * behind this GPU's caches, so no line of it may live in them. mij_output_is_uncached tells (1 / 0). */
MIJ_API int mij_encoder_reserve_output(mij_encoder *enc, size_t scan_capacity_bytes);
MIJ_API int mij_output_is_uncached(const mij_encoder *enc);
MIJ_API int mij_output_buffer(mij_encoder *enc, void **d_buffer, size_t *scan_offset, size_t *scan_capacity);
#define MIJ_IPC_HANDLE_BYTES 64
MIJ_API int mij_ipc_export(const void *d_ptr, void *handle64);
MIJ_API int mij_ipc_open(int device, const void *handle64, void **d_ptr);
MIJ_API int mij_ipc_close(void *d_ptr);

/* Waits for `stream` work issued by the last encode and reports where the bitstream is. */
MIJ_API int mij_encode_result(mij_encoder *enc, mij_result *out);

/* nvjpegEncodeRetrieveBitstream (ImageCompressorImpl.cu:285-287), same two-call protocol: data == NULL -> *length
 * receives the size; otherwise up to *length bytes are copied to host memory and *length is updated. */
MIJ_API int mij_retrieve_bitstream(mij_encoder *enc, uint8_t *data, size_t *length);

/* CompressWorker's marshalling (ImageCompressorImpl.cu:272-277) + encode + retrieve in one call, from host memory:
 * uploads the interleaved / planar image once (no cv::split) in ranges of MCU rows, running the transform stage of
 * each range while the next one is on the wire, encodes, and returns a library-owned page-locked host buffer that
 * stays valid until the next call on this handle. Fastest when `src` is page-locked (mij_host_alloc). */
MIJ_API int mij_encode_host(mij_encoder *enc, const uint8_t *src, size_t pitch, size_t plane_stride, int input_format,
                            const uint8_t **jpeg, size_t *jpeg_bytes);

/* Page-locked host memory for images handed to mij_encode_host / mij_decode_host (replaces the pageable cv::Mat data of
 * the reference, ImageCompressorImpl.cu:273-277, whose three cudaMemcpy calls stage through the driver). */
MIJ_API int mij_host_alloc(void **ptr, size_t bytes);
MIJ_API void mij_host_free(void *ptr);

/* Per-stage device times of the last encode in milliseconds (the reference prints one cudaEvent time,
 * ImageCompressorImpl.cu:289-291): [0] transform [1] statistics [2] table build [3] entropy code [4] scan
 * [5] stuff+compact [6] total. Requires mij_encoder_enable_timing(enc, 1) before the encode. (Progressive encoders report
 * the ten scans together under [2] and zeros for [3]..[5].) */
#define MIJ_NUM_STAGE_TIMES 7
MIJ_API int mij_encoder_enable_timing(mij_encoder *enc, int on);
MIJ_API int mij_stage_times(mij_encoder *enc, float ms[MIJ_NUM_STAGE_TIMES]);

/* Debug / parity taps (device -> host copies of intermediate buffers; used by the parity tests). */
MIJ_API int mij_debug_coefficients(mij_encoder *enc, int16_t *host_dst, size_t count);
MIJ_API int mij_debug_tables(mij_encoder *enc, uint8_t *host_dst_4x273);

/* ------------------------------------------------------------------------------------------------------------------
 * Decode (reference initDecodeEnv / DecodeWorker, ImageCompressorImpl.cu:67-117, 311-385). Accepted: 8-bit Huffman-coded
 * JPEG, baseline / extended sequential (SOF0, SOF1) and progressive (SOF2, the mode the reference's encoder writes),
 * 3 components (1x1 chroma; luma 1x1, 2x1, 2x2, 1x2, 4x1, 4x2) or greyscale, with or without restart markers.
 *  - One interleaved sequential scan (this library's output, camera files): decoded in parallel -- every restart interval
 *    is further cut into 1-KiB subsequences that are decoded speculatively and synchronised (k_decode_par.inc), so a file
 *    WITHOUT restart markers is as fast as one with them.
 *  - Progressive and multi-scan files WITH restart markers: exact, one GPU lane per restart interval of each scan
 *    (k_decode_scans.inc).
 *  - Progressive files WITHOUT restart markers (what the reference's encoder writes, ImageCompressorImpl.cu:256-259): each
 *    scan in parallel (k_decode_prog.inc). DC-first and AC-first scans synchronise like a sequential scan (the state of the
 *    decoder is the bit position and the place in the MCU / the EOB run). An AC refinement scan does not synchronise by
 *    itself -- how many correction bits a block takes depends on which of its coefficients are already non-zero -- so block
 *    positions are FOUND: hypotheses "block b starts at bit p" are walked against the history maps and die on a
 *    violation, survivors that agree unanimously become anchors, and the scan is then decoded exactly from anchor to anchor,
 *    each segment having to arrive on the next anchor bit-exactly. Where the survivors do not agree (thin histories: the
 *    paths a whole number of blocks beside the true one meet no violation either) they stay on as candidates and the state
 *    the anchor before ARRIVES at picks the true one. A scan where all this finds nothing to hold on to (the thinnest scans of
 *    large smooth images, scans of a few hundred bytes) is walked by one WAVE instead (k_decode_wave.inc: exact as well, orders of magnitude slower:
 *    seconds for hundreds of megapixels); mij_decode_px_report says which scans went which way.
 * Pixels are identical to libjpeg-turbo's (islow IDCT, fancy upsampling). */
typedef struct mij_decoder mij_decoder;
/* initDecodeEnv (ImageCompressorImpl.cu:67-95) / destoryDecodeEnv (.cu:97-117). NULL destroy is a no-op. */
MIJ_API int mij_decoder_create(int device, mij_decoder **out);
MIJ_API void mij_decoder_destroy(mij_decoder *dec);
MIJ_API const char *mij_decoder_last_error(const mij_decoder *dec);
/* nvjpegGetImageInfo (ImageCompressorImpl.cu:335): host-side parse only; css uses the MIJ_CSS_* values. */
MIJ_API int mij_decode_info(const uint8_t *jpeg, size_t jpeg_bytes, int *width, int *height, int *css, int *restart_interval);
/* nvjpegJpegStreamParse + DecodeJpegHost + TransferToDevice + DecodeJpegDevice (ImageCompressorImpl.cu:362-366) with the
 * planar->interleaved step of getCVImageOnCPU (.cu:214-221) done on the device: host JPEG bytes -> device pixels.
 * output_format: MIJ_INPUT_BGRI / RGBI (interleaved, pitch >= 3*width) or MIJ_INPUT_BGR / RGB (planar, 3 planes at
 * plane_stride; the reference's NVJPEG_OUTPUT_BGR, ImageCompressorImpl.cuh:69). `jpeg` may also point to DEVICE memory
 * (e.g. mij_result.d_buffer + header_offset): the entropy-coded data is then decoded in place. Returns once the entropy decoding is
 * synchronised (its passes are counted from the host); the remaining kernels are asynchronous on `stream`.
 * mij_decode_sync waits and reports stream errors and the device time in ms. */
MIJ_API int mij_decode_device(mij_decoder *dec, const uint8_t *jpeg, size_t jpeg_bytes, void *d_dst, size_t pitch,
                              size_t plane_stride, int output_format, void *stream);
MIJ_API int mij_decode_sync(mij_decoder *dec, float *device_ms);
/* Device time in ms of the last completed decode on this handle (what the reference means to print as
 * "=> Decode Cost time", ImageCompressorImpl.cu:368-373, where the start event is never recorded), and the handle's device. */
MIJ_API int mij_decode_last_ms(const mij_decoder *dec, float *device_ms);
MIJ_API int mij_decoder_device(const mij_decoder *dec);
/* Progressive files without restart markers (round 4): how many scans of the last decode the parallel decoder (first scans by
 * subsequence synchronisation, AC refinement scans by hypothesis search + exact verification) was tried on, and how many of those
 * it decoded; the rest were walked by one wave each (exact as well, orders of magnitude slower). Waits for the decode. */
MIJ_API int mij_decode_px_report(mij_decoder *dec, int *scans_tried, int *scans_parallel);
/* Footprint of that parallel decoder: its workspace is sized by the FILE and kept in the handle -- per AC refinement scan two hypothesis
 * lists of 16 bytes x max(8 M, what the scan's anchors can ask for; at most 2^28) entries plus ~60 bytes per block: about 0.3 GB for a
 * 1920x1080 file, ~2.5 GB per luma refinement scan of 8320x40000, all scans of a file together. It is never a reason for a decode to fail:
 * above MIJ_PX_WS_BUDGET_MB (environment, default 24576 = 24 GiB; 0 = never use the parallel decoder's workspace) or when the device
 * refuses the allocation, the largest scans are walked by one wave each instead (exact, no workspace, slower) until the rest fits.
 * MIJ_PROG_PARALLEL=0 turns the parallel decoder off altogether. */
/* DecodeWorker end to end (ImageCompressorImpl.cu:311-385): host JPEG bytes -> host pixels (one D2H, already interleaved). */
MIJ_API int mij_decode_host(mij_decoder *dec, const uint8_t *jpeg, size_t jpeg_bytes, uint8_t *dst, size_t pitch,
                            int output_format, int *width, int *height);

/* Secondary ("difference map") compression, reference README.md:8 (no code in the reference; definition in SURVEY.md 8a A9):
 * mode -1: out = clip(a - b + 128)  (residual of original a against decoded b)
 * mode +1: out = clip(a + b - 128)  (reconstruction from decoded a and decoded residual b).  n = number of bytes. */
MIJ_API int mij_residual_device(const void *d_a, const void *d_b, void *d_out, size_t n, int mode, void *stream);

/* The first layer's reconstruction WITHOUT decoding its file: after mij_encode_transform / mij_encode_device of image I on this
 * handle, D = dec(enc(I)) is a function of the quantised coefficients the handle still holds (entropy coding is lossless), so
 * this call runs dequantisation + inverse DCT + upsampling + colour conversion straight from them -- pixel for pixel what
 * mij_decode_device makes of the handle's file -- and stores, asynchronously on `stream`,
 *   d_src != NULL:  R = clip(I - D + 128)   (the difference map; d_src = the image the coefficients came from, same format)
 *   d_src == NULL:  D itself.
 * `input_format` (MIJ_INPUT_*) describes d_src AND d_dst. Whole images only. Valid until the handle's next transform. */
MIJ_API int mij_encode_residual_device(mij_encoder *enc, const void *d_src, size_t pitch, size_t plane_stride, int input_format,
                                       void *d_dst, size_t dst_pitch, size_t dst_plane_stride, void *stream);

/* The whole two-layer scheme in one call each way (host memory in, host memory out; whole images, not strips):
 *   encode:  J1 = enc(I);  D = dec(J1);  R = clip(I - D + 128);  J2 = enc(R)          (same quality / sampling for both layers)
 *   decode:  I' = clip(dec(J1) + dec(J2) - 128)
 * `primary` / `secondary` are caller buffers; *primary_bytes / *secondary_bytes hold their capacities on entry and the
 * file sizes on return. MIJ_ERR_OVERFLOW if either is too small: *primary_bytes then holds the size the first layer needs,
 * *secondary_bytes the size the second layer needs (0 if the first layer already did not fit and the second was not coded);
 * grow the buffers and call again. A residual image is close to noise: at high quality with dense restart markers a layer
 * can exceed the raw image size. The encode side no longer needs the decoder (D comes from the encoder's coefficients,
 * mij_encode_residual_device): `dec` may be NULL; if given it must be on the encoder's device. */
MIJ_API int mij_secondary_encode_host(mij_encoder *enc, mij_decoder *dec, const uint8_t *src, size_t pitch, size_t plane_stride,
                                      int input_format, uint8_t *primary, size_t *primary_bytes, uint8_t *secondary,
                                      size_t *secondary_bytes);
MIJ_API int mij_secondary_decode_host(mij_decoder *dec, const uint8_t *primary, size_t primary_bytes, const uint8_t *secondary,
                                      size_t secondary_bytes, uint8_t *dst, size_t pitch, int output_format, int *width, int *height);

/* The second layer with parameters of its own (round 4; reference README.md:8 names the scheme in one sentence and no code, so
 * the definition is this library's): at the first layer's quality and sampling the difference map is below
 * the quantiser's step and a second layer adds bytes but hardly any fidelity; coded finer, at full chroma resolution and / or
 * amplified, it does.
 *   encode:  J1 = enc(I; encoder's quality, css);  D = dec(J1);  R = clip((I - D) * gain + 128);  J2 = enc(R; quality2, css2)
 *   decode:  I' = clip(dec(J1) + floor((dec(J2) - 128 + gain / 2) / gain))
 * Defaults (quality2 = 0, css2 = -1, gain = 0 or 1; or params == NULL) reproduce mij_secondary_encode_host / _decode_host
 * bit for bit. The second layer is coded by a second encoder handle kept inside `enc` (created on first use, re-created when
 * quality2 / css2 change, destroyed with `enc`), with the restart interval MIJ_RESTART_AUTO picks for its sampling. Each
 * layer is an ordinary JFIF file: byte-identical to what a stock encoder makes of I / of R at that layer's settings. The
 * gain is not stored in either file: the caller passes the same params to the decode side (only `gain` is read there).
 * Footprint: that second handle owns a complete set of encoder workspaces of its own (coefficients 2(1+f) bytes per pixel, entropy
 * scratch and output buffer: roughly 4 GB for 8320x40000 at 4:4:4), kept until `enc` is destroyed or the parameters change; with
 * quality2 / css2 left at the first layer's values no second handle exists and nothing extra is held. If it cannot be created the call
 * fails with mij_encoder_create's own code and message (MIJ_ERR_ALLOC: "... hipMalloc ..."). Both layers are coded on one stream,
 * without a host wait between them. */
typedef struct mij_secondary_params {
  uint32_t struct_size;   /* = sizeof(mij_secondary_params) */
  int quality2;           /* 1..100; 0 = the first layer's */
  int css2;               /* MIJ_CSS_*; -1 = the first layer's */
  int gain;               /* 1, 2, 4 or 8; 0 = 1 */
} mij_secondary_params;
#define MIJ_SECONDARY_PARAMS_INIT {(uint32_t)sizeof(mij_secondary_params), 0, -1, 1}
MIJ_API int mij_secondary_encode_host_ex(mij_encoder *enc, const mij_secondary_params *params, const uint8_t *src, size_t pitch,
                                         size_t plane_stride, int input_format, uint8_t *primary, size_t *primary_bytes,
                                         uint8_t *secondary, size_t *secondary_bytes);
MIJ_API int mij_secondary_decode_host_ex(mij_decoder *dec, const mij_secondary_params *params, const uint8_t *primary, size_t primary_bytes,
                                         const uint8_t *secondary, size_t secondary_bytes, uint8_t *dst, size_t pitch, int output_format,
                                         int *width, int *height);
/* mij_residual_device with the gain above: mode -1: out = clip((a - b) * gain + 128); mode +1: out = clip(a + floor((b - 128 + gain / 2) / gain)). */
MIJ_API int mij_residual_gain_device(const void *d_a, const void *d_b, void *d_out, size_t n, int mode, int gain, void *stream);
/* mij_encode_residual_device with the gain above: d_dst <- clip((I - D) * gain + 128) (d_src == NULL: D itself, gain ignored). */
MIJ_API int mij_encode_residual_gain_device(mij_encoder *enc, const void *d_src, size_t pitch, size_t plane_stride, int input_format,
                                            void *d_dst, size_t dst_pitch, size_t dst_plane_stride, int gain, void *stream);

/* Progressive output in STRIPS (round 5): the reference's own encoding (ImageCompressorImpl.cu:28) sharded like the baseline path.
 * A progressive encoder may be created for a strip of MCU rows when the restart interval divides the MCUs per row and the width is a
 * whole number of MCUs: a strip is then a whole number of restart intervals in every one of the ten scans, and the only image-wide
 * quantity is each scan's symbol statistics. Per image, on every rank:
 *   mij_encode_transform(enc, ...);
 *   mij_encode_prog_statistics(enc, stream);             the ten scans' counts -> one device buffer (asynchronous on `stream`)
 *   [caller: all-reduce (sum) of mij_prog_histogram_buffer over the ranks, on `stream`]
 *   mij_encode_prog_emit(enc, stream, sizes, header_bytes);   tables from the image-wide counts, this strip's restart intervals coded;
 *                                                         waits; sizes[i] = bytes of the strip's segment of scan i (every interval with
 *                                                         its RSTn; the last strip's segments end without one), header_bytes[i] = bytes in
 *                                                         front of scan i's data in the file (SOI .. SOS for i = 0, else DHT + SOS):
 *                                                         identical on every rank
 *   [caller: all-gather of sizes; offset of (scan i, rank r) = sum_{j <= i} header_bytes[j] + sum_{j < i, all r'} sizes[r'][j]
 *                                                              + sum_{r' < r} sizes[r'][i];  file_bytes = the last such end + 2]
 *   mij_encode_prog_place(enc, offsets, d_file, capacity, file_bytes, flags, stream);
 *                                                         this strip's ten segments to d_file + offsets[i] (device memory of this process:
 *                                                         the assembling rank's file buffer, or a staging buffer the caller then sends from;
 *                                                         NULL = the handle's own buffer, grown as needed, after which mij_encode_result /
 *                                                         mij_retrieve_bitstream return the file). Up to 2 bytes behind a segment may be
 *                                                         overwritten (the last interval's dropped marker): leave that gap when staging.
 *                                                         flags: MIJ_PROG_PLACE_HEADERS = also write the ten headers in front of offsets[i]
 *                                                         (the rank of the FIRST strip, whose offsets are those of rank 0),
 *                                                         MIJ_PROG_PLACE_EOI = also write the EOI at file_bytes - 2. Waits for its copies.
 * One rank with the whole image gets, through these calls, the file mij_encode_entropy writes. */
#define MIJ_PROG_SCANS 10
#define MIJ_PROG_PLACE_HEADERS 1
#define MIJ_PROG_PLACE_EOI 2
MIJ_API int mij_encode_prog_statistics(mij_encoder *enc, void *stream);
MIJ_API int mij_prog_histogram_buffer(mij_encoder *enc, void **d_ptr, size_t *words);      /* uint32 words, 10 x 4 x 257 */
MIJ_API int mij_encode_prog_emit(mij_encoder *enc, void *stream, uint64_t sizes[MIJ_PROG_SCANS], uint64_t header_bytes[MIJ_PROG_SCANS]);
MIJ_API int mij_encode_prog_place(mij_encoder *enc, const uint64_t offsets[MIJ_PROG_SCANS], void *d_file, size_t file_capacity,
                                  uint64_t file_bytes, int flags, void *stream);

/* Bench utility: fill device memory with rows [y0, y0+rows) of the SURVEY.md 8(d) synthetic image
 * (RGB or BGR interleaved). */
/* Bench utility: the streaming-copy yardstick (16 B per lane, 4 loads in flight): read + write `bytes` each; all three
 * arguments 16-byte aligned. */
MIJ_API int mij_copy_bench_device(void *d_dst, const void *d_src, size_t bytes, void *stream);
/* Bench utility: the shader clock the SIMDs actually run at (the encode kernels are bound by vector instruction issue, so a
 * throughput figure is only comparable between boxes together with it). One launch of a fixed vector-ALU loop at 8 waves per
 * SIMD: *valu_clock_mhz = issued 4-cycle instructions per SIMD x 4 / launch duration; *counter_clock_mhz (may be NULL) = the
 * ratio of the shader-clock counter to the constant-rate counter inside that launch x hipDeviceAttributeWallClockRate (0 if
 * that rate is unknown); *launch_ms (may be NULL) = the launch duration. iters = 1024 runs ~0.9 ms at 2.4 GHz. Synchronous. */
MIJ_API int mij_clock_probe_device(int iters, void *stream, double *valu_clock_mhz, double *counter_clock_mhz, double *launch_ms);
MIJ_API int mij_synth_image_device(void *d_dst, int width, int y0, int rows, size_t pitch, int bgr, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MI_JPEG_H_ */

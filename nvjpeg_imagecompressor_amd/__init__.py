"""MI355X-native JPEG path behind the NvjpegCompressRunner surface (drop-in for the nvJPEG hot path of
OroChippw/Nvjpeg-ImageCompressor). HIP kernels + C ABI live in csrc/ and libmijpeg.so; this package is host glue."""
from ._lib import CSS, MIJ_RESTART_AUTO, MiJpegError, LIB_PATH  # noqa: F401
from .encoder import (Decoder, Encoder, NvjpegCompressRunner, clock_probe_device, copy_bench_device, geometry_query, library_source_hash,  # noqa: F401
                      pinned_empty, residual_device, synth_image_device)

__all__ = ["Encoder", "Decoder", "NvjpegCompressRunner", "synth_image_device", "residual_device", "pinned_empty", "CSS", "MIJ_RESTART_AUTO", "MiJpegError"]

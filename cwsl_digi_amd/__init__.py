"""cwsl_digi_amd -- MI355X-native replacement for CWSL_DIGI's per-Instance DSP chain.

The product is the HIP shared library cwsl_digi_amd/lib/libcwslgpu.so (C ABI: include/cwsl_gpu.h).
This package is only the loader plus a thin Python mirror of that ABI for tests and bench.py.
There is no CPU fallback anywhere in this package.
"""
from .api import (Context, CwslGpuError, GROUPS, group_of, frame_len, load_library,  # noqa: F401
                  STATUS_NAMES, parse_decoder_line, decoder_block_bytes, decoder_block_field,
                  decoder_route, decoder_command, slot_clock_next, pool_sizing, find_band, parse_decode_line, rccl_unique_id, exact_stream_length)

__all__ = ["Context", "CwslGpuError", "GROUPS", "group_of", "frame_len", "load_library", "STATUS_NAMES", "parse_decoder_line",
           "decoder_block_bytes", "decoder_block_field", "decoder_route", "decoder_command",
           "slot_clock_next", "pool_sizing", "find_band", "parse_decode_line", "rccl_unique_id", "exact_stream_length"]

"""ctypes mirror of include/cwsl_gpu.h.

Naming follows the reference's seams (source/Receiver.hpp, source/Instance.cpp, source/SSBD.hpp):
a Context owns Receivers (IQ rings in HBM) and Channels (one SSBD + frame pair each); slot
boundaries are signalled per SyncPredicates group.  Every call goes through libcwslgpu.so; if the
library or a gfx950 device is missing this raises -- nothing here computes DSP on the CPU.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
# CWSLG_LIB: another build of the library -- a path, or "lab" for lib/libcwslgpu_lab.so (the measured alternative kernels and the
# environment switches that select them; the product library has neither)
_LIB_PATH = os.environ.get("CWSLG_LIB") or os.path.join(_HERE, "lib", "libcwslgpu.so")
if _LIB_PATH == "lab":
    _LIB_PATH = os.path.join(_HERE, "lib", "libcwslgpu_lab.so")

STATUS_NAMES = {
    0: "OK", -1: "ERR_RATIO", -2: "ERR_BAND_LOW", -3: "ERR_BAND_HIGH", -4: "ERR_NOMEM", -5: "ERR_MODE",
    -6: "ERR_ARG", -7: "ERR_NO_DEVICE", -8: "ERR_HIP", -9: "ERR_NO_FRAME", -10: "ERR_UNSUPPORTED",
    -11: "ERR_BLOCK",
}
ERR_NO_FRAME = -9
ERR_HIP = -8

# CWSL_DIGI_Types.hpp:83-143
GROUPS = {"FT8": 0, "FT4": 1, "Q65_30": 2, "S60": 3, "S120": 4, "S300": 5, "S900": 6, "S1800": 7}
_MODE_GROUP = {
    "FT8": 0, "JS8": 0, "FT4": 1, "Q65-30": 2, "JT65": 3, "FST4-60": 3, "WSPR": 4, "FST4-120": 4,
    "FST4W-120": 4, "FST4-300": 5, "FST4W-300": 5, "FST4-900": 6, "FST4W-900": 6, "FST4-1800": 7,
    "FST4W-1800": 7,
}
_MODE_PERIOD = {
    "FT8": 15.0, "JS8": 15.0, "FT4": 7.5, "WSPR": 120.0, "Q65-30": 30.0, "JT65": 60.0, "FST4-60": 60.0,
    "FST4-120": 120.0, "FST4-300": 300.0, "FST4-900": 900.0, "FST4-1800": 1800.0, "FST4W-120": 120.0,
    "FST4W-300": 300.0, "FST4W-900": 900.0, "FST4W-1800": 1800.0,
}


def group_of(mode):
    return _MODE_GROUP[mode]


def frame_len(mode):
    """Instance.cpp:149 : 12000 * (period + 5)."""
    return int(12000.0 * float(np.float32(_MODE_PERIOD[mode]) + np.float32(5)))


class CwslGpuError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        name = STATUS_NAMES.get(status, str(status))
        super().__init__(f"libcwslgpu: {name}" + (f": {detail}" if detail else ""))


class DecoderSpec(C.Structure):
    _fields_ = [("freq_hz", C.c_uint32), ("calibrated_hz", C.c_uint32), ("mode", C.c_char * 16), ("smnum", C.c_int32),
                ("freqcal", C.c_double), ("callsign", C.c_char * 16), ("group", C.c_int32),
                ("frame_len", C.c_uint32), ("period_s", C.c_float)]


def parse_decoder_line(line, freqcal_global=1.0):
    """config.ini `decoder=` value -> dict (CWSL_DIGI.cpp:731-837).  Raises CwslGpuError on the reference's error cases."""
    L = load_library()
    sp = DecoderSpec()
    rc = L.cwslg_parse_decoder_line(line.encode(), float(freqcal_global), C.byref(sp))
    if rc != 0:
        raise CwslGpuError(rc, "Error parsing decoder line: " + line)
    return dict(freq_hz=sp.freq_hz, calibrated_hz=sp.calibrated_hz, mode=sp.mode.decode(), smnum=sp.smnum,
                freqcal=sp.freqcal, callsign=sp.callsign.decode(), group=sp.group, frame_len=sp.frame_len,
                period_s=sp.period_s)


class Candidate(C.Structure):
    _fields_ = [("freq_bin", C.c_int32), ("time_step", C.c_int32), ("sync", C.c_float),
                ("freq_hz", C.c_float), ("dt_s", C.c_float)]


class Spot(C.Structure):
    _fields_ = [("snr_db", C.c_int32), ("dt_s", C.c_float), ("freq_hz", C.c_uint32), ("has_locator", C.c_int32),
                ("call", C.c_char * 16), ("locator", C.c_char * 8), ("message", C.c_char * 64), ("drift", C.c_int32), ("dbm", C.c_int32)]


class Ft4Sync(C.Structure):
    _fields_ = [("f0_hz", C.c_float), ("f1_hz", C.c_float), ("dt_s", C.c_float), ("sync", C.c_float),
                ("ibest", C.c_int32), ("idf", C.c_int32), ("seg", C.c_int32), ("cand", C.c_int32)]


class WsprCandidate(C.Structure):
    _fields_ = [("freq_hz", C.c_float), ("snr_db", C.c_float), ("drift", C.c_float), ("sync", C.c_float), ("shift", C.c_int32)]


class Fst4wCandidate(C.Structure):
    _fields_ = [("freq_hz", C.c_float), ("snr", C.c_float), ("bin", C.c_int32), ("pad_", C.c_int32)]


class SlotResult(C.Structure):
    _fields_ = [("start_epoch", C.c_uint64), ("n_valid", C.c_uint64), ("factor", C.c_float), ("list_kind", C.c_int32),
                ("n_list", C.c_int32), ("n_ft4_sync", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("demod_launches", C.c_uint64), ("demod_samples", C.c_uint64),
                ("finalize_launches", C.c_uint64), ("sync_launches", C.c_uint64), ("frames_emitted", C.c_uint64),
                ("frames_discarded", C.c_uint64), ("blocks_dropped", C.c_uint64),
                ("h2d_bytes", C.c_uint64), ("demod_ms", C.c_double), ("finalize_ms", C.c_double),
                ("sync_ms", C.c_double), ("phasor_regrows", C.c_uint64), ("rendezvous_calls", C.c_uint64),
                ("rendezvous_frames", C.c_uint64), ("rccl_world", C.c_uint64), ("rendezvous_flags_and", C.c_uint64),
                ("demod_clock_mhz", C.c_double), ("demod_clock_launches", C.c_uint64), ("push_calls", C.c_uint64),
                ("push_batches", C.c_uint64), ("push_host_ms", C.c_double), ("sync_spectra_ms", C.c_double), ("sync_search_ms", C.c_double),
                ("demod_blocks_read", C.c_uint64), ("process_deferred", C.c_uint64)]


# int (*)(void *user, int group, uint64_t epoch_s, uint64_t frames_local, uint64_t *frames_total)
RENDEZVOUS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64))

_lib = None

# every symbol include/cwsl_gpu.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = [
    "cwslg_abi_version", "cwslg_create", "cwslg_destroy", "cwslg_strerror", "cwslg_last_error",
    "cwslg_set_scale_factors", "cwslg_set_exact", "cwslg_receiver_open", "cwslg_receiver_close", "cwslg_push_iq", "cwslg_push_iq_many",
    "cwslg_push_iq_device", "cwslg_push_synth", "cwslg_ring_commit", "cwslg_ring_commit_all", "cwslg_ring_info", "cwslg_parse_decoder_line", "cwslg_channel_open_line", "cwslg_channel_open", "cwslg_channel_close", "cwslg_channel_tune", "cwslg_channel_tune_ex",
    "cwslg_channel_info", "cwslg_process", "cwslg_set_process_threshold", "cwslg_flush", "cwslg_exact_stream_length", "cwslg_slot_boundary", "cwslg_slot_boundary_begin", "cwslg_slot_boundary_end", "cwslg_slot_boundary_channel",
    "cwslg_set_boundary_rendezvous", "cwslg_set_rendezvous_flag", "cwslg_rccl_unique_id", "cwslg_rccl_init",
    "cwslg_enable_long_sync", "cwslg_fetch_wspr_candidates", "cwslg_fetch_fst4w_candidates", "cwslg_long_sync_debug_fetch",
    "cwslg_synchronize", "cwslg_fetch_frame", "cwslg_fetch_slot", "cwslg_write_wav", "cwslg_fetch_audio_f32", "cwslg_frame_device_ptrs",
    "cwslg_enable_sync", "cwslg_set_candidate_order", "cwslg_fetch_candidates", "cwslg_set_ft4_syncmin", "cwslg_enable_ft4_coherent", "cwslg_fetch_ft4_sync", "cwslg_sync_debug_fetch", "cwslg_get_stats", "cwslg_reset_stats",
    "cwslg_set_timing", "cwslg_demod_kernel_name", "cwslg_stream", "cwslg_channel_constants", "cwslg_phasor_checkpoint_stride", "cwslg_channel_phasor_checkpoints",
    "cwslg_slot_clock_next", "cwslg_pool_sizing", "cwslg_find_band", "cwslg_parse_decode_line",
    "cwslg_decoder_block_bytes", "cwslg_decoder_block_field", "cwslg_fill_decoder_block", "cwslg_decoder_route", "cwslg_decoder_command",
]


def rccl_unique_id():
    """128-byte ncclUniqueId made by this process (rank 0 hands it to the other ranks)."""
    L = load_library()
    buf = C.create_string_buffer(128)
    rc = L.cwslg_rccl_unique_id(buf)
    if rc < 0:
        raise CwslGpuError(rc, L.cwslg_strerror(rc).decode())
    return buf.raw


def load_library(build_if_missing=True):
    """dlopen libcwslgpu.so.  Raises (never falls back) if it is absent and cannot be built."""
    global _lib
    if _lib is not None:
        return _lib
    if not build_if_missing and not os.path.isfile(_LIB_PATH):
        raise CwslGpuError(-7, f"{_LIB_PATH} not built")
    if build_if_missing and os.environ.get("CWSLG_LIB", "lab") == "lab":
        _build.build()          # returns at once when the library is newer than every source; serialised by a file lock
    try:
        # If torch is (or will be) in the process, let it load ITS libamdhip64 first: both copies carry
        # SONAME libamdhip64.so.7 and two HIP runtimes in one process do not share device pointers.
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    L.cwslg_abi_version.restype = i32
    L.cwslg_create.argtypes = [C.POINTER(vp), i32]
    L.cwslg_destroy.argtypes = [vp]; L.cwslg_destroy.restype = None
    L.cwslg_strerror.argtypes = [i32]; L.cwslg_strerror.restype = C.c_char_p
    L.cwslg_last_error.argtypes = [vp]; L.cwslg_last_error.restype = C.c_char_p
    L.cwslg_set_scale_factors.argtypes = [vp, f32, f32]
    L.cwslg_set_exact.argtypes = [vp, i32]
    L.cwslg_demod_kernel_name.argtypes = [vp]; L.cwslg_demod_kernel_name.restype = C.c_char_p
    L.cwslg_set_boundary_rendezvous.argtypes = [vp, RENDEZVOUS_FN, vp]
    L.cwslg_set_rendezvous_flag.argtypes = [vp, u64]
    L.cwslg_rccl_unique_id.argtypes = [vp]
    L.cwslg_rccl_init.argtypes = [vp, vp, i32, i32]
    L.cwslg_enable_long_sync.argtypes = [vp, i32, i32, i32, f32]
    L.cwslg_fetch_wspr_candidates.argtypes = [vp, i32, vp, i32, C.POINTER(i32), C.POINTER(u64)]
    L.cwslg_fetch_fst4w_candidates.argtypes = [vp, i32, vp, i32, C.POINTER(i32), C.POINTER(u64)]
    L.cwslg_fetch_slot.argtypes = [vp, i32, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(Ft4Sync), i32, C.POINTER(SlotResult)]
    L.cwslg_long_sync_debug_fetch.argtypes = [vp, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cwslg_receiver_open.argtypes = [vp, u32, u32, C.c_int32, u32, C.POINTER(i32)]
    L.cwslg_receiver_close.argtypes = [vp, i32]
    L.cwslg_push_iq.argtypes = [vp, i32, vp, u32]
    L.cwslg_push_iq_many.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(vp), u32]
    L.cwslg_push_iq_device.argtypes = [vp, i32, vp, u32]
    L.cwslg_push_synth.argtypes = [vp, i32, u64, u32, u32, vp, i32, f32]
    L.cwslg_ring_commit.argtypes = [vp, i32, u32, u32]
    L.cwslg_ring_commit_all.argtypes = [vp, u32, u32]
    L.cwslg_ring_info.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(u32), C.POINTER(u64)]
    L.cwslg_parse_decoder_line.argtypes = [C.c_char_p, C.c_double, C.POINTER(DecoderSpec)]
    L.cwslg_channel_open_line.argtypes = [vp, i32, C.c_char_p, C.c_double, C.POINTER(i32)]
    L.cwslg_channel_open.argtypes = [vp, i32, C.c_int32, i32, C.c_char_p, C.POINTER(i32)]
    L.cwslg_channel_close.argtypes = [vp, i32]
    L.cwslg_channel_tune.argtypes = [vp, i32, C.c_int32, i32]
    L.cwslg_channel_tune_ex.argtypes = [vp, i32, C.c_int32, i32, i32]
    L.cwslg_channel_info.argtypes = [vp, i32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(u32), C.POINTER(C.c_size_t)]
    L.cwslg_process.argtypes = [vp]
    L.cwslg_set_process_threshold.argtypes = [vp, i32]
    L.cwslg_flush.argtypes = [vp]
    L.cwslg_exact_stream_length.argtypes = [u64, u32, u32, i32]; L.cwslg_exact_stream_length.restype = u32
    L.cwslg_slot_boundary.argtypes = [vp, i32, u64]
    L.cwslg_slot_boundary_begin.argtypes = [vp, i32, u64]
    L.cwslg_slot_boundary_end.argtypes = [vp]
    L.cwslg_slot_boundary_channel.argtypes = [vp, i32, u64]
    L.cwslg_synchronize.argtypes = [vp]
    L.cwslg_fetch_frame.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(u64), C.POINTER(C.c_size_t), C.POINTER(f32)]
    L.cwslg_write_wav.argtypes = [vp, i32, C.c_char_p]
    L.cwslg_fetch_audio_f32.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cwslg_frame_device_ptrs.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp)]
    L.cwslg_enable_sync.argtypes = [vp, i32, f32, i32, i32, i32]
    L.cwslg_fetch_candidates.argtypes = [vp, i32, C.POINTER(Candidate), i32, C.POINTER(i32), C.POINTER(u64)]
    L.cwslg_set_candidate_order.argtypes = [vp, i32]
    L.cwslg_set_ft4_syncmin.argtypes = [vp, f32]
    L.cwslg_enable_ft4_coherent.argtypes = [vp, i32]
    L.cwslg_fetch_ft4_sync.argtypes = [vp, i32, C.POINTER(Ft4Sync), i32, C.POINTER(i32), C.POINTER(u64)]
    L.cwslg_sync_debug_fetch.argtypes = [vp, i32, i32, vp, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(i32)]
    L.cwslg_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.cwslg_reset_stats.argtypes = [vp]
    L.cwslg_set_timing.argtypes = [vp, i32]
    L.cwslg_stream.argtypes = [vp]; L.cwslg_stream.restype = vp
    L.cwslg_channel_constants.argtypes = [vp, i32, vp, vp, vp]
    L.cwslg_channel_phasor_checkpoints.argtypes = [vp, i32, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cwslg_decoder_block_bytes.argtypes = [i32]; L.cwslg_decoder_block_bytes.restype = C.c_size_t
    L.cwslg_decoder_block_field.argtypes = [i32, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.cwslg_fill_decoder_block.argtypes = [vp, i32, vp, C.c_size_t, i32, i32, i32, C.POINTER(u64)]
    L.cwslg_decoder_route.argtypes = [C.c_char_p, i32]
    L.cwslg_decoder_command.argtypes = [C.c_char_p, i32, i32, i32, i32, i32, f32, C.c_char_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.cwslg_parse_decode_line.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.POINTER(Spot)]
    L.cwslg_slot_clock_next.argtypes = [i32, u64]; L.cwslg_slot_clock_next.restype = u64
    L.cwslg_pool_sizing.argtypes = [C.POINTER(i32), f32, i32, C.POINTER(i32), C.POINTER(i32)]
    L.cwslg_find_band.argtypes = [C.POINTER(C.c_int64), C.POINTER(u32), i32, C.c_int64]
    _lib = L
    return L


def parse_decode_line(mode, line, base_freq_hz):
    """One jt9 stdout line -> dict(status 'ok'|'unhandled'|'skip', snr_db, dt_s, freq_hz, call, locator, message)
    (OutputHandler.cpp:505-621, 924-1128)."""
    sp = Spot()
    rc = load_library().cwslg_parse_decode_line(mode.encode(), line.encode(), int(base_freq_hz), C.byref(sp))
    if rc < 0:
        raise CwslGpuError(rc, f"parse_decode_line({mode})")
    return dict(status=("ok", "unhandled", "skip")[rc], snr_db=sp.snr_db, dt_s=sp.dt_s, freq_hz=sp.freq_hz,
                call=sp.call.decode(), locator=sp.locator.decode() if sp.has_locator else None, message=sp.message.decode(), drift=sp.drift, dbm=sp.dbm)


def slot_clock_next(group, after_ms):
    """UTC ms of the group's (or mode's) next slot boundary strictly after after_ms (CWSL_DIGI.cpp:174-451)."""
    g = (GROUPS[group] if group in GROUPS else _MODE_GROUP[group]) if isinstance(group, str) else int(group)
    return int(load_library().cwslg_slot_clock_next(g, int(after_ms)))


POOL_ORDER = ("FT4", "FT8", "Q65-30", "JS8", "WSPR", "JT65", "FST4W", "FST4")


def pool_sizing(counts, decoderburden=1.0, n_decoders=None):
    """(numjt9instances, maxwsprdinstances) from decoder counts in POOL_ORDER (CWSL_DIGI.cpp:857-887)."""
    arr = (C.c_int * 8)(*[int(x) for x in counts])
    nj, nw = C.c_int(), C.c_int()
    rc = load_library().cwslg_pool_sizing(arr, float(decoderburden), int(sum(counts) if n_decoders is None else n_decoders),
                                          C.byref(nj), C.byref(nw))
    if rc != 0:
        raise CwslGpuError(rc, "pool_sizing")
    return nj.value, nw.value


def exact_stream_length(total_blocks, max_blocks, cu_count=256, latency=True):
    """cwslg_exact_stream_length: outputs per stream of the bit-identical kernel for a launch of that size (pure; no GPU needed)."""
    return int(load_library().cwslg_exact_stream_length(int(total_blocks), int(max_blocks), int(cu_count), 1 if latency else 0))


def find_band(bands, f_hz):
    """bands = [(lo_hz, fs_hz), ...]; index of the first band covering f_hz or -1 (CWSL_Utils.hpp:28-55)."""
    lo = (C.c_int64 * len(bands))(*[int(b[0]) for b in bands])
    fs = (C.c_uint32 * len(bands))(*[int(b[1]) for b in bands])
    return int(load_library().cwslg_find_band(lo, fs, len(bands), int(f_hz)))


def decoder_block_bytes(js8=False):
    """sizeof(dec_data_t) / sizeof(dec_data_js8_t) (DecoderPool.hpp:58-108 / :110-171)."""
    return load_library().cwslg_decoder_block_bytes(int(js8))


def decoder_block_field(name, js8=False):
    """(offset, bytes) of a member of the decoder block: ipc/ss/savg/sred/d2/params or a params field name."""
    off, n = C.c_size_t(), C.c_size_t()
    rc = load_library().cwslg_decoder_block_field(int(js8), name.encode(), C.byref(off), C.byref(n))
    if rc != 0:
        raise CwslGpuError(rc, f"no member {name!r} in the decoder block")
    return off.value, n.value


def decoder_route(mode, transfer_shmem=True):
    """'shmem' or 'wavefile' -- DecoderPool.hpp:379-395."""
    rc = load_library().cwslg_decoder_route(mode.encode(), int(transfer_shmem))
    if rc < 0:
        raise CwslGpuError(rc, f"Unhandled mode: {mode}")
    return "shmem" if rc == 1 else "wavefile"


def decoder_command(mode, target, shmem_route, numjt9threads=3, decodedepth=3, highest_decode_hz=3000, wspr_cycles=3000,
                    trperiod=0.0):
    """(program, argument string) exactly as DecoderPool.hpp:634-659 / :1007-1046 concatenate them."""
    app, opts = C.create_string_buffer(64), C.create_string_buffer(1024)
    rc = load_library().cwslg_decoder_command(mode.encode(), int(shmem_route), numjt9threads, decodedepth, highest_decode_hz,
                                              wspr_cycles, float(trperiod), str(target).encode(), app, 64, opts, 1024)
    if rc != 0:
        raise CwslGpuError(rc, f"Mode {mode} not handled")
    return app.value.decode(), opts.value.decode()


class Context:
    """One GPU's worth of receivers and channels (one process per GPU)."""

    def __init__(self, device=-1):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.cwslg_create(C.byref(h), device)
        if rc != 0:
            raise CwslGpuError(rc, self.L.cwslg_strerror(rc).decode())
        self.h = h
        self._modes = {}

    def close(self):
        if getattr(self, "h", None):
            self.L.cwslg_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc < 0:
            raise CwslGpuError(rc, self.L.cwslg_last_error(self.h).decode() or self.L.cwslg_strerror(rc).decode())
        return rc

    # ---- Receiver.hpp ----
    def receiver_open(self, fs=192000, iq_len=2048, lo_hz=0, ring_blocks=0):
        rid = C.c_int(-1)
        self._chk(self.L.cwslg_receiver_open(self.h, fs, iq_len, lo_hz, ring_blocks, C.byref(rid)))
        return rid.value

    def receiver_close(self, rx):
        self._chk(self.L.cwslg_receiver_close(self.h, rx))

    def push_iq(self, rx, iq):
        """iq: complex64[n] host array (one or more Receiver blocks)."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        self._chk(self.L.cwslg_push_iq(self.h, rx, iq.ctypes.data, iq.shape[0]))

    def push_iq_many(self, rxs, blocks):
        """One block per receiver in ONE call (cwslg_push_iq_many): rxs = receiver ids, blocks = complex64 arrays of one common length
        (or one 2-D array, row k for rxs[k])."""
        arrs = [np.ascontiguousarray(b, dtype=np.complex64) for b in blocks]
        n = arrs[0].shape[0]
        assert len(arrs) == len(rxs) and all(a.shape == (n,) for a in arrs)
        ids = (C.c_int * len(rxs))(*[int(r) for r in rxs])
        ptrs = (C.c_void_p * len(rxs))(*[a.ctypes.data for a in arrs])
        self._chk(self.L.cwslg_push_iq_many(self.h, len(rxs), ids, ptrs, n))

    def push_iq_device(self, rx, dptr, n_complex):
        self._chk(self.L.cwslg_push_iq_device(self.h, rx, C.c_void_p(dptr), n_complex))

    def push_synth(self, rx, seed, n_complex, block_len=0, tones_hz=(), amp=2.0e4):
        t = np.ascontiguousarray(tones_hz, dtype=np.float64)
        self._chk(self.L.cwslg_push_synth(self.h, rx, seed, n_complex, block_len,
                                          t.ctypes.data if len(t) else None, len(t), amp))

    def ring_commit(self, rx, n_complex, block_len=0):
        self._chk(self.L.cwslg_ring_commit(self.h, rx, n_complex, block_len))

    def ring_commit_all(self, n_complex, block_len=0):
        self._chk(self.L.cwslg_ring_commit_all(self.h, n_complex, block_len))

    def ring_info(self, rx):
        p, cap, tot = C.c_void_p(), C.c_uint32(), C.c_uint64()
        self._chk(self.L.cwslg_ring_info(self.h, rx, C.byref(p), C.byref(cap), C.byref(tot)))
        return p.value, cap.value, tot.value

    # ---- Instance.cpp / SSBD.hpp ----
    def channel_open(self, rx, demod_hz, mode="FT8", usb=True):
        cid = C.c_int(-1)
        self._chk(self.L.cwslg_channel_open(self.h, rx, int(demod_hz), 1 if usb else 0, mode.encode(), C.byref(cid)))
        self._modes[cid.value] = mode
        return cid.value

    def channel_open_line(self, rx, line, freqcal_global=1.0):
        """Open a channel from an unchanged config.ini `decoder=` value."""
        cid = C.c_int(-1)
        self._chk(self.L.cwslg_channel_open_line(self.h, rx, line.encode(), float(freqcal_global), C.byref(cid)))
        self._modes[cid.value] = parse_decoder_line(line, freqcal_global)["mode"]
        return cid.value

    def channel_close(self, ch):
        self._chk(self.L.cwslg_channel_close(self.h, ch))
        self._modes.pop(ch, None)

    def channel_tune(self, ch, demod_hz, usb=True, reset=True):
        """SSBD::Tune(F, isUSB, reset): retune in place; reset=True restarts history and phasor, reset=False keeps them
        (SSBD.hpp:96-123)."""
        self._chk(self.L.cwslg_channel_tune_ex(self.h, ch, int(demod_hz), 1 if usb else 0, 1 if reset else 0))

    def channel_info(self, ch):
        a, b, c_, d = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        fl = C.c_size_t()
        self._chk(self.L.cwslg_channel_info(self.h, ch, C.byref(a), C.byref(b), C.byref(c_), C.byref(d), C.byref(fl)))
        return dict(in_size=a.value, out_size=b.value, out_rate=c_.value, delay=d.value, frame_len=fl.value)

    def process(self):
        self._chk(self.L.cwslg_process(self.h))

    def flush(self):
        """cwslg_flush: demodulate everything pending now, whatever the process threshold."""
        self._chk(self.L.cwslg_flush(self.h))

    def set_process_threshold(self, min_outputs=-1):
        """cwslg_set_process_threshold: 0 every process() launches; > 0 only once a channel has that many outputs pending; < 0 the library's own."""
        self._chk(self.L.cwslg_set_process_threshold(self.h, int(min_outputs)))

    def slot_boundary(self, group, epoch_s):
        if isinstance(group, str):                       # a group name ("S120") or a mode name ("WSPR" -> its group)
            g = GROUPS[group] if group in GROUPS else _MODE_GROUP[group]
        else:
            g = int(group)
        self._chk(self.L.cwslg_slot_boundary(self.h, g, int(epoch_s)))

    def slot_boundary_begin(self, group, epoch_s):
        """First half of slot_boundary: queue the boundary's device work, return at once (see cwslg_slot_boundary_begin)."""
        g = (GROUPS[group] if group in GROUPS else _MODE_GROUP[group]) if isinstance(group, str) else int(group)
        self._chk(self.L.cwslg_slot_boundary_begin(self.h, g, int(epoch_s)))

    def slot_boundary_end(self):
        """Second half: wait for that boundary's kernels and run the rendezvous."""
        self._chk(self.L.cwslg_slot_boundary_end(self.h))

    def set_boundary_rendezvous(self, fn):
        """Install the multi-GPU slot-boundary rendezvous: fn(group, epoch_s, frames_local) -> frames over all
        processes (e.g. shard.slot_boundary_rendezvous over torch.distributed).  None removes it."""
        if fn is None:
            self._rdv_cb = None
            self._chk(self.L.cwslg_set_boundary_rendezvous(self.h, RENDEZVOUS_FN(), None))
            return

        def tramp(_user, group, epoch_s, frames_local, total_out):
            try:
                total_out[0] = int(fn(int(group), int(epoch_s), int(frames_local)))
                return 0
            except Exception:                              # an exception must not unwind through the C caller
                import traceback
                traceback.print_exc()
                return ERR_HIP
        self._rdv_cb = RENDEZVOUS_FN(tramp)                 # keep the trampoline alive as long as it is installed
        self._chk(self.L.cwslg_set_boundary_rendezvous(self.h, self._rdv_cb, None))

    def set_rendezvous_flag(self, flag):
        """The 64-bit word this process contributes to every later built-in rendezvous (stats()["rendezvous_flags_and"])."""
        self._chk(self.L.cwslg_set_rendezvous_flag(self.h, int(flag)))

    def rccl_init(self, unique_id, rank, world):
        """Built-in RCCL rendezvous (what a C++ host uses): unique_id = bytes from rccl_unique_id() of rank 0."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._chk(self.L.cwslg_rccl_init(self.h, buf, int(rank), int(world)))

    def slot_boundary_channel(self, ch, epoch_s):
        self._chk(self.L.cwslg_slot_boundary_channel(self.h, ch, int(epoch_s)))

    def synchronize(self):
        self._chk(self.L.cwslg_synchronize(self.h))

    def fetch_frame(self, ch):
        """-> None until a frame was finalised, else dict(i16, t_start, n_valid, factor)."""
        n = frame_len(self._modes[ch])
        out = np.empty(n, np.int16)
        t0, nv, fac = C.c_uint64(), C.c_size_t(), C.c_float()
        rc = self.L.cwslg_fetch_frame(self.h, ch, out.ctypes.data, n, C.byref(t0), C.byref(nv), C.byref(fac))
        if rc == ERR_NO_FRAME:
            return None
        self._chk(rc)
        return dict(i16=out, t_start=t0.value, n_valid=nv.value, factor=np.float32(fac.value))

    def write_wav(self, ch, path):
        """The reference's 46-byte-header 12 kHz mono int16 .wav of the last finalised frame (WaveFile.hpp:87-135)."""
        self._chk(self.L.cwslg_write_wav(self.h, ch, str(path).encode()))

    def fill_decoder_block(self, ch, block=None, js8=False, decodedepth=3, highest_decode_hz=3000):
        """The jt9/js8 shared-memory block for the last finalised frame (DecoderPool.hpp:451-590).  `block` is a
        writable uint8 array of decoder_block_bytes(js8) (e.g. a mapped shared-memory segment) or None to allocate.
        Returns (block, t_start) or None before the first frame."""
        n = decoder_block_bytes(js8)
        if block is None:
            block = np.empty(n, np.uint8)
        t0 = C.c_uint64()
        rc = self.L.cwslg_fill_decoder_block(self.h, ch, block.ctypes.data, block.nbytes, int(js8), decodedepth,
                                             highest_decode_hz, C.byref(t0))
        if rc == ERR_NO_FRAME:
            return None
        self._chk(rc)
        return block, t0.value

    def fetch_audio_f32(self, ch):
        n = frame_len(self._modes[ch])
        out = np.empty(n, np.float32)
        nv = C.c_size_t()
        rc = self.L.cwslg_fetch_audio_f32(self.h, ch, out.ctypes.data, n, C.byref(nv))
        if rc == ERR_NO_FRAME:
            return None
        self._chk(rc)
        return out, nv.value

    def frame_device_ptrs(self, ch):
        a, b = C.c_void_p(), C.c_void_p()
        self._chk(self.L.cwslg_frame_device_ptrs(self.h, ch, C.byref(a), C.byref(b)))
        return a.value, b.value

    def enable_sync(self, enable=True, syncmin=1.5, max_cand=200, f_lo_hz=200, f_hi_hz=3000):
        self._chk(self.L.cwslg_enable_sync(self.h, 1 if enable else 0, syncmin, max_cand, f_lo_hz, f_hi_hz))

    def set_candidate_order(self, order="sync"):
        """'sync' (default): strongest first, cut at max_cand in that order; 'freq': ascending frequency, cut in THAT order."""
        self._chk(self.L.cwslg_set_candidate_order(self.h, {"sync": 0, "freq": 1}[order]))

    def fetch_candidates(self, ch, max_cand=600, with_epoch=False):
        """[(freq_bin, time_step, sync, freq_hz, dt_s)]; with_epoch: (list, start epoch of the frame the list was computed from)."""
        buf = (Candidate * max_cand)()
        n = C.c_int()
        t0 = C.c_uint64()
        self._chk(self.L.cwslg_fetch_candidates(self.h, ch, buf, max_cand, C.byref(n), C.byref(t0)))
        out = [(buf[k].freq_bin, buf[k].time_step, buf[k].sync, buf[k].freq_hz, buf[k].dt_s) for k in range(n.value)]
        return (out, t0.value) if with_epoch else out

    def fetch_slot(self, ch, max_list=1000, max_ft4=3000, want_frame=True):
        """cwslg_fetch_slot: frame + candidate list(s) of ONE epoch under one ticket.  -> None before the first frame, else a dict with
        i16 (or None), t_start, n_valid, factor, list_kind ('none' / 'FT8' / 'FT4' / 'WSPR' / 'FST4W'), list (tuples in the layout of the
        matching fetch_*_candidates call) and ft4_sync (dicts as fetch_ft4_sync)."""
        n = frame_len(self._modes[ch])
        out = np.empty(n, np.int16) if want_frame else None
        item = max(C.sizeof(Candidate), C.sizeof(WsprCandidate), C.sizeof(Fst4wCandidate))
        raw = (C.c_char * (item * max_list))()
        ft4 = (Ft4Sync * max_ft4)()
        res = SlotResult()
        rc = self.L.cwslg_fetch_slot(self.h, ch, out.ctypes.data if want_frame else None, n if want_frame else 0, raw, len(raw), ft4, max_ft4, C.byref(res))
        if rc == ERR_NO_FRAME:
            return None
        self._chk(rc)
        kind = ("none", "FT8", "FT4", "WSPR", "FST4W")[res.list_kind]
        lst = []
        if kind in ("FT8", "FT4"):
            b = C.cast(raw, C.POINTER(Candidate))
            lst = [(b[k].freq_bin, b[k].time_step, b[k].sync, b[k].freq_hz, b[k].dt_s) for k in range(res.n_list)]
        elif kind == "WSPR":
            b = C.cast(raw, C.POINTER(WsprCandidate))
            lst = [(b[k].freq_hz, b[k].snr_db, b[k].drift, b[k].sync, b[k].shift) for k in range(res.n_list)]
        elif kind == "FST4W":
            b = C.cast(raw, C.POINTER(Fst4wCandidate))
            lst = [(b[k].freq_hz, b[k].snr, b[k].bin) for k in range(res.n_list)]
        f4 = [dict(f0_hz=b.f0_hz, f1_hz=b.f1_hz, dt_s=b.dt_s, sync=b.sync, ibest=b.ibest, idf=b.idf, seg=b.seg, cand=b.cand) for b in ft4[:res.n_ft4_sync]]
        return dict(i16=out, t_start=res.start_epoch, n_valid=res.n_valid, factor=res.factor, list_kind=kind, list=lst, ft4_sync=f4)

    def enable_ft4_coherent(self, enable=True):
        self._chk(self.L.cwslg_enable_ft4_coherent(self.h, int(enable)))

    def fetch_ft4_sync(self, ch, max_rec=1800):
        """Refined FT4 candidates (ft4_downsample + sync4d search): list of dicts, candidate order then segment order."""
        buf = (Ft4Sync * max_rec)()
        n = C.c_int()
        rc = self.L.cwslg_fetch_ft4_sync(self.h, ch, buf, max_rec, C.byref(n), None)
        if rc == ERR_NO_FRAME:
            return None
        self._chk(rc)
        return [dict(f0_hz=b.f0_hz, f1_hz=b.f1_hz, dt_s=b.dt_s, sync=b.sync, ibest=b.ibest, idf=b.idf, seg=b.seg, cand=b.cand)
                for b in buf[:n.value]]

    def enable_long_sync(self, on=True, nfa_hz=1400, nfb_hz=1600, minsync=1.2):
        """Candidate search of the 120 s modes (WSPR: wsprd's front end; FST4W-120: get_candidates_fst4 over nfa..nfb)."""
        self._chk(self.L.cwslg_enable_long_sync(self.h, 1 if on else 0, nfa_hz, nfb_hz, minsync))

    def fetch_wspr_candidates(self, ch, max_cand=200, with_epoch=False):
        """-> None until a frame was searched, else [(freq_hz, snr_db, drift, sync, shift)] (with_epoch: (list, frame start epoch))."""
        buf = (WsprCandidate * max_cand)()
        n = C.c_int()
        t0 = C.c_uint64()
        rc = self.L.cwslg_fetch_wspr_candidates(self.h, ch, buf, max_cand, C.byref(n), C.byref(t0))
        if rc == ERR_NO_FRAME:
            return None
        self._chk(rc)
        out = [(b.freq_hz, b.snr_db, b.drift, b.sync, b.shift) for b in buf[:n.value]]
        return (out, t0.value) if with_epoch else out

    def fetch_fst4w_candidates(self, ch, max_cand=100, with_epoch=False):
        """-> None until a frame was searched, else [(freq_hz, snr, bin)] (with_epoch: (list, frame start epoch))."""
        buf = (Fst4wCandidate * max_cand)()
        n = C.c_int()
        t0 = C.c_uint64()
        rc = self.L.cwslg_fetch_fst4w_candidates(self.h, ch, buf, max_cand, C.byref(n), C.byref(t0))
        if rc == ERR_NO_FRAME:
            return None
        self._chk(rc)
        out = [(b.freq_hz, b.snr, b.bin) for b in buf[:n.value]]
        return (out, t0.value) if with_epoch else out

    def long_sync_debug(self, ch, what):
        """what: 'iq' complex64[46080], 'ps' float32[512, 359] (wsprd's ps[j][i]), 'smspec' float32[411] (WSPR);
        's2' float32[nnw], 'band' complex64[nband] (FST4W)."""
        sel = {"iq": 0, "ps": 1, "smspec": 2, "s2": 3, "band": 4}[what]
        buf = np.empty(2 * 46080 if sel == 0 else (359 * 512 if sel == 1 else (411 if sel == 2 else 65600)), np.float32)
        n = C.c_size_t()
        self._chk(self.L.cwslg_long_sync_debug_fetch(self.h, ch, sel, buf.ctypes.data, buf.nbytes, C.byref(n)))
        out = buf[:n.value]
        if sel in (0, 4):
            return out.view(np.complex64)
        if sel == 1:
            return np.ascontiguousarray(out.reshape(359, 512).T)
        return out

    def set_ft4_syncmin(self, syncmin=1.2):
        self._chk(self.L.cwslg_set_ft4_syncmin(self.h, syncmin))

    def sync_debug(self, ch, what):
        """what: 'spectra' -> float32[steps, row]; 'red'/'red2' -> float32[1921]; 'jpeak'/'jpeak2' -> int32[1921].
        FT4 channels: 'red' = savsm/sbase, 'red2' = sbase (first 1153 entries meaningful)."""
        sel = {"spectra": 0, "red": 1, "red2": 2, "jpeak": 3, "jpeak2": 4, "ft4_cx": 5, "ft4_cd0": 6}[what]
        if sel >= 5:                                    # FT4 coherent stage: complex64[36289] / complex64[4032]
            buf = np.empty(2 * (36289 if sel == 5 else 4032), np.float32)
            n, row = C.c_size_t(), C.c_int()
            self._chk(self.L.cwslg_sync_debug_fetch(self.h, ch, sel, buf.ctypes.data, buf.nbytes, C.byref(n), C.byref(row)))
            return buf.view(np.complex64)
        buf = np.empty(372 * 1936 if sel == 0 else 1921, np.float32 if sel < 3 else np.int32)
        n, row = C.c_size_t(), C.c_int()
        self._chk(self.L.cwslg_sync_debug_fetch(self.h, ch, sel, buf.ctypes.data, buf.nbytes, C.byref(n), C.byref(row)))
        buf = buf[: n.value]
        return buf.reshape(-1, row.value) if sel == 0 else buf

    def stats(self):
        s = Stats()
        self._chk(self.L.cwslg_get_stats(self.h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in Stats._fields_}

    def reset_stats(self):
        self._chk(self.L.cwslg_reset_stats(self.h))

    def set_timing(self, on=True):
        self._chk(self.L.cwslg_set_timing(self.h, 1 if on else 0))

    def set_scale_factors(self, ft=0.90, wspr=0.20):
        self._chk(self.L.cwslg_set_scale_factors(self.h, ft, wspr))

    def set_exact(self, on=True):
        """Reference-order arithmetic: float and int16 frames bit-identical to the compiled reference."""
        self._chk(self.L.cwslg_set_exact(self.h, 1 if on else 0))

    def demod_kernel_name(self):
        """Name of the kernel the most recent demod launch ran (bench.py's roofline.kernel)."""
        return (self.L.cwslg_demod_kernel_name(self.h) or b"").decode()

    def stream(self):
        return self.L.cwslg_stream(self.h)

    def channel_constants(self, ch):
        """(taps float32[32D], tone complex64[D], phase_inc complex64) exactly as uploaded."""
        D = self.channel_info(ch)["in_size"] // 4
        taps = np.empty(32 * D, np.float32)
        tone = np.empty(2 * D, np.float32)
        inc = np.empty(2, np.float32)
        self._chk(self.L.cwslg_channel_constants(self.h, ch, taps.ctypes.data, tone.ctypes.data, inc.ctypes.data))
        return taps, tone.view(np.complex64), inc.view(np.complex64)[0]

    def checkpoint_stride(self):
        return int(self.L.cwslg_phasor_checkpoint_stride())

    def phasor_checkpoints(self, ch, n=None):
        tot = C.c_size_t()
        self._chk(self.L.cwslg_channel_phasor_checkpoints(self.h, ch, None, 0, C.byref(tot)))
        n = tot.value if n is None else min(n, tot.value)
        out = np.empty(2 * n, np.float32)
        self._chk(self.L.cwslg_channel_phasor_checkpoints(self.h, ch, out.ctypes.data, n, C.byref(tot)))
        return out.view(np.complex64)

"""TEST INFRASTRUCTURE ONLY.

ctypes bindings for
  * oracle/liboracle.so       -- this repo's plain-C restatement (cwsl_oracle.c, sync_oracle.c)
  * oracle/_ref/libcwsl_ref.so -- the unmodified reference SSBD.hpp/LowPass.hpp compiled in place
                                 (present only where `make -C oracle ref` could run)

Nothing in the product path (cwsl_digi_amd/) may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i16p = np.ctypeslib.ndpointer(np.int16, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile liboracle.so (always) and _ref/libcwsl_ref.so (when the reference is present)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if ref and os.path.isfile("/root/reference/source/SSBD.hpp"):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


class _Demod(C.Structure):
    _fields_ = [
        ("fs", C.c_uint64), ("bw", C.c_uint64), ("usb", C.c_int), ("sign", C.c_float),
        ("block", C.c_uint32), ("ntaps", C.c_uint32),
        ("taps", C.POINTER(C.c_float)), ("tone_re", C.POINTER(C.c_float)), ("tone_im", C.POINTER(C.c_float)),
        ("inc_re", C.c_float), ("inc_im", C.c_float), ("ph_re", C.c_float), ("ph_im", C.c_float),
        ("ws_re", C.c_float * 32), ("ws_im", C.c_float * 32),
        ("head", C.c_uint32), ("phase_delta", C.c_float),
    ]


class _Channel(C.Structure):
    _fields_ = [
        ("mode", C.c_char * 16), ("fs", C.c_uint64), ("iq_len", C.c_uint32), ("demod_hz", C.c_int32),
        ("scale_ft", C.c_float), ("scale_wspr", C.c_float), ("frame_len", C.c_size_t),
        ("frame", C.POINTER(C.c_float) * 2), ("fill", C.c_uint64 * 2), ("t0", C.c_uint64 * 2),
        ("wr", C.c_uint32), ("rd", C.c_uint32), ("demod", _Demod), ("dropped_blocks", C.c_uint64),
    ]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        # CWSL_ORACLE_LIB: another build of the same sources, e.g. `make -C oracle liboracle_asan.so` run with
        # LD_PRELOAD=$(gcc -print-file-name=libasan.so) -- the sanitizer pass over the checker itself
        path = os.environ.get("CWSL_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
        if not os.path.isfile(path):
            build(ref=False)
        L = C.CDLL(path)
        L.orc_lowpass_design.argtypes = [C.c_size_t, C.c_double, _f32p]
        L.orc_demod_open.argtypes = [C.POINTER(_Demod), C.c_uint64, C.c_uint64, C.c_double, C.c_int]
        L.orc_demod_close.argtypes = [C.POINTER(_Demod)]
        L.orc_demod_tune.argtypes = [C.POINTER(_Demod), C.c_double, C.c_int]
        L.orc_demod_tune_ex.argtypes = [C.POINTER(_Demod), C.c_double, C.c_int, C.c_int]
        L.orc_demod_run.argtypes = [C.POINTER(_Demod), _f32p, C.c_uint64, _f32p, C.c_void_p]
        L.orc_rx_period.argtypes = [C.c_char_p]; L.orc_rx_period.restype = C.c_double
        L.orc_frame_len.argtypes = [C.c_char_p]; L.orc_frame_len.restype = C.c_size_t
        L.orc_prepare_audio.argtypes = [_f32p, C.c_size_t, C.c_char_p, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_prepare_audio.restype = C.c_float
        L.orc_to_int16.argtypes = [_f32p, C.c_size_t, _i16p]
        L.orc_channel_open.argtypes = [C.POINTER(_Channel), C.c_char_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_float, C.c_float]
        L.orc_channel_close.argtypes = [C.POINTER(_Channel)]
        L.orc_channel_push.argtypes = [C.POINTER(_Channel), _f32p]
        L.orc_channel_boundary.argtypes = [C.POINTER(_Channel), C.c_uint64, _i16p, C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_float)]
        L.orc_wav_header.argtypes = [C.c_uint32, C.c_char_p]
        L.orc_wav_write.argtypes = [C.c_char_p, _i16p, C.c_uint32]
        L.orc_mix64.argtypes = [C.c_uint64]; L.orc_mix64.restype = C.c_uint64
        L.orc_synth_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f32p]
        L.orc_synth_add_tones.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, _f64p, C.c_int, C.c_float, _f32p]
        L.orc_checksum_f32.argtypes = [_f32p, C.c_size_t]; L.orc_checksum_f32.restype = C.c_double
        L.orc_ft8_spectra.argtypes = [_i16p, _f32p, C.c_int]
        L.orc_ft8_sync.argtypes = [_i16p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ft8_sync_ordered.argtypes = [_i16p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ft4_candidates_ordered.argtypes = [_i16p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_ft4_spectra.argtypes = [_i16p, _f32p]
        L.orc_ft4_candidates.argtypes = [_i16p, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_log10_fixed.argtypes = [C.c_double]; L.orc_log10_fixed.restype = C.c_double
        L.orc_exp10_fixed.argtypes = [C.c_double]; L.orc_exp10_fixed.restype = C.c_double
        L.orc_bench_cpu.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint64]
        L.orc_bench_cpu.restype = C.c_double
        L.orc_crc32.argtypes = [C.c_void_p, C.c_size_t]; L.orc_crc32.restype = C.c_uint32
        L.orc_jt9_block_bytes.restype = C.c_size_t
        L.orc_js8_block_bytes.restype = C.c_size_t
        L.orc_jt9_offset.argtypes = [C.c_char_p]; L.orc_jt9_offset.restype = C.c_long
        L.orc_js8_offset.argtypes = [C.c_char_p]; L.orc_js8_offset.restype = C.c_long
        L.orc_jt9_fill.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, _i16p, C.c_size_t]
        L.orc_js8_fill.argtypes = [C.c_void_p, C.c_int, C.c_int, _i16p, C.c_size_t]
        L.orc_decoder_route.argtypes = [C.c_char_p, C.c_int]
        L.orc_decoder_command.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_char_p,
                                          C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_ft4_bigspec.argtypes = [_i16p, _f32p]
        L.orc_ft4_downsample.argtypes = [_f32p, C.c_float, _f32p]
        L.orc_ft4_sync4d.argtypes = [_f32p, C.c_int, C.c_int]; L.orc_ft4_sync4d.restype = C.c_float
        L.orc_ft4_search.argtypes = [_f32p, C.c_float, C.c_int, C.c_void_p, C.c_int]
        L.orc_parse_decode_line.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.c_void_p]
        L.orc_clock_sim.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_pool_sizing.argtypes = [C.POINTER(C.c_int), C.c_float, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.orc_find_band.argtypes = [C.POINTER(C.c_int64), C.POINTER(C.c_uint32), C.c_int, C.c_int64]
        _lib = L
    return _lib


def have_ref():
    return os.path.isfile(os.path.join(_HERE, "_ref", "libcwsl_ref.so"))


def ref():
    """The compiled reference headers.  Raises if the build container did not produce them."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libcwsl_ref.so")
        if not os.path.isfile(path):
            raise RuntimeError("oracle/_ref/libcwsl_ref.so absent (run `make -C oracle ref` where /root/reference exists)")
        R = C.CDLL(path)
        R.ref_ssbd_new.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_int, C.c_char_p, C.c_int]
        R.ref_ssbd_new.restype = C.c_void_p
        R.ref_ssbd_delete.argtypes = [C.c_void_p]
        for n in ("in_size", "out_size", "out_rate", "delay", "filt_order", "block_size"):
            f = getattr(R, "ref_ssbd_" + n); f.argtypes = [C.c_void_p]; f.restype = C.c_uint64
        R.ref_ssbd_get_taps.argtypes = [C.c_void_p, _f32p]
        R.ref_ssbd_get_tone.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        R.ref_ssbd_run.argtypes = [C.c_void_p, _f32p, C.c_uint64, _f32p, C.c_void_p]
        R.ref_ssbd_run.restype = C.c_int
        R.ref_build_lowpass.argtypes = [C.c_uint64, C.c_double, _f32p]
        R.ref_ssbd_tune.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_char_p, C.c_int]
        R.ref_ssbd_tune_ex.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_char_p, C.c_int]
        R.ref_bench_cpu.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint64]
        R.ref_bench_cpu.restype = C.c_double
        R.ref_inst_new.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_uint64]; R.ref_inst_new.restype = C.c_void_p
        R.ref_inst_delete.argtypes = [C.c_void_p]
        R.ref_inst_push.argtypes = [C.c_void_p, _f32p]
        R.ref_inst_boundary.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        R.ref_mode_group.argtypes = [C.c_char_p]
        R.ref_is_valid_locator.argtypes = [C.c_char_p]
        R.ref_trim.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        _ref = R
    return _ref


# ----------------------------------------------------------------------------------------------
# numpy-level helpers
# ----------------------------------------------------------------------------------------------
ERRORS = {-1: "Fs/B must be an even integer >= 4", -2: "Signal outside of band (low)",
          -3: "Signal outside of band (high)", -4: "alloc", -5: "Unhandled mode"}


class Demod:
    """orc_demod_t: the SSBD<float> restatement."""

    def __init__(self, fs, f_hz, bw=6000, usb=True):
        self.s = _Demod()
        rc = lib().orc_demod_open(C.byref(self.s), fs, bw, float(f_hz), 1 if usb else 0)
        if rc != 0:
            raise ValueError(ERRORS.get(rc, str(rc)))
        self.block = self.s.block
        self.ntaps = self.s.ntaps

    def close(self):
        if self.s is not None:
            lib().orc_demod_close(C.byref(self.s)); self.s = None

    __del__ = close

    @property
    def taps(self):
        return np.ctypeslib.as_array(self.s.taps, (self.ntaps,)).copy()

    @property
    def tone(self):
        re = np.ctypeslib.as_array(self.s.tone_re, (self.block,))
        im = np.ctypeslib.as_array(self.s.tone_im, (self.block,))
        t = np.empty(self.block, np.complex64)   # no arithmetic: keep -0.0 signs
        t.real = re; t.imag = im
        return t

    @property
    def phase_inc(self):
        return np.array([self.s.inc_re, self.s.inc_im], np.float32).view(np.complex64)[0]

    @property
    def phase_delta(self):
        return np.float32(self.s.phase_delta)

    def tune(self, f_hz, usb=True, reset=True):
        """SSBD::Tune(F, isUSB, reset) on the live object; raises ValueError with the reference's text, state untouched."""
        rc = lib().orc_demod_tune_ex(C.byref(self.s), float(f_hz), 1 if usb else 0, 1 if reset else 0)
        if rc != 0:
            raise ValueError({-2: "Signal outside of band (low)", -3: "Signal outside of band (high)"}.get(rc, str(rc)))

    def run(self, iq, trace=False):
        """iq: complex64[n] (n multiple of 4*block) -> float32[n/block] (+ complex64 phasor trace)."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        n = iq.shape[0]
        assert n % (4 * self.block) == 0
        out = np.empty(n // self.block, np.float32)
        tr = np.empty(2 * (n // self.block), np.float32) if trace else None
        lib().orc_demod_run(C.byref(self.s), iq.view(np.float32), n, out,
                            tr.ctypes.data if trace else None)
        return (out, tr.view(np.complex64)) if trace else out


class RefInstance:
    """Instance's frame handling replayed on the reference's own ring_buffer_t / sample_buffer_t / SSBD (oracle/_ref)."""

    def __init__(self, mode, fs, iq_len, demod_hz):
        self.iq_len, self.frame_len = iq_len, frame_len(mode)
        self.h = ref().ref_inst_new(fs, iq_len, float(demod_hz), self.frame_len)
        if not self.h:
            raise ValueError("ref_inst_new failed")

    def close(self):
        if getattr(self, "h", None):
            ref().ref_inst_delete(self.h); self.h = None

    __del__ = close

    def push(self, iq_block):
        iq_block = np.ascontiguousarray(iq_block, dtype=np.complex64)
        assert iq_block.shape[0] == self.iq_len
        return int(ref().ref_inst_push(self.h, iq_block.view(np.float32)))

    def boundary(self, epoch_s):
        """-> (emitted, t_start, samples_written, float32 frame before prepareAudio or None)."""
        f32 = np.empty(self.frame_len, np.float32)
        t0, nw = C.c_uint64(0), C.c_uint64(0)
        got = ref().ref_inst_boundary(self.h, int(epoch_s), f32.ctypes.data, C.byref(t0), C.byref(nw))
        return bool(got), int(t0.value), int(nw.value), (f32 if got else None)


def ref_mode_group(mode):
    return int(ref().ref_mode_group(mode.encode()))


def ref_is_valid_locator(s):
    return bool(ref().ref_is_valid_locator(s.encode()))


def ref_trim(s):
    out = C.create_string_buffer(len(s) + 8)
    ref().ref_trim(s.encode(), out, len(s) + 8)
    return out.value.decode()


class RefDemod:
    """SSBD<float> itself (oracle/_ref)."""

    def __init__(self, fs, f_hz, bw=6000, usb=True):
        err = C.create_string_buffer(256)
        self.h = ref().ref_ssbd_new(fs, bw, float(f_hz), 1 if usb else 0, err, 256)
        if not self.h:
            raise ValueError(err.value.decode())
        self.block = int(ref().ref_ssbd_block_size(self.h))
        self.ntaps = int(ref().ref_ssbd_filt_order(self.h))

    def close(self):
        if getattr(self, "h", None):
            ref().ref_ssbd_delete(self.h); self.h = None

    __del__ = close

    def tune(self, f_hz, usb=True, reset=True):
        err = C.create_string_buffer(256)
        if ref().ref_ssbd_tune_ex(self.h, float(f_hz), 1 if usb else 0, 1 if reset else 0, err, 256) != 0:
            raise ValueError(err.value.decode())

    @property
    def taps(self):
        t = np.empty(self.ntaps, np.float32); ref().ref_ssbd_get_taps(self.h, t); return t

    def _tone(self):
        t = np.empty(2 * self.block, np.float32); inc = np.empty(2, np.float32); ph = np.empty(2, np.float32)
        ref().ref_ssbd_get_tone(self.h, t, inc, ph)
        return t.view(np.complex64), inc.view(np.complex64)[0], ph.view(np.complex64)[0]

    @property
    def tone(self):
        return self._tone()[0]

    @property
    def phase_inc(self):
        return self._tone()[1]

    def run(self, iq, trace=False):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        n = iq.shape[0]
        assert n % (4 * self.block) == 0
        out = np.empty(n // self.block, np.float32)
        tr = np.empty(2 * (n // self.block), np.float32) if trace else None
        rc = ref().ref_ssbd_run(self.h, iq.view(np.float32), n, out, tr.ctypes.data if trace else None)
        assert rc == 0
        return (out, tr.view(np.complex64)) if trace else out


class Channel:
    """orc_channel_t: Instance::sampleManager restated (one channel, 2-frame ring)."""

    def __init__(self, mode, fs, iq_len, demod_hz, scale_ft=0.90, scale_wspr=0.20):
        self.c = _Channel()
        rc = lib().orc_channel_open(C.byref(self.c), mode.encode(), fs, iq_len, int(demod_hz), scale_ft, scale_wspr)
        if rc != 0:
            raise ValueError(ERRORS.get(rc, str(rc)))
        self.frame_len = self.c.frame_len
        self.iq_len = iq_len

    def close(self):
        if self.c is not None:
            lib().orc_channel_close(C.byref(self.c)); self.c = None

    __del__ = close

    def tune(self, demod_hz, usb=True, reset=True):
        """Instance's SSBD retuned in place (SSBD::Tune): float(demod_hz) as at construction (Instance.cpp:187)."""
        rc = lib().orc_demod_tune_ex(C.byref(self.c.demod), float(np.float32(demod_hz)), 1 if usb else 0, 1 if reset else 0)
        if rc != 0:
            raise ValueError({-2: "Signal outside of band (low)", -3: "Signal outside of band (high)"}.get(rc, str(rc)))

    def push(self, iq_block):
        iq_block = np.ascontiguousarray(iq_block, dtype=np.complex64)
        assert iq_block.shape[0] == self.iq_len
        return lib().orc_channel_push(C.byref(self.c), iq_block.view(np.float32))

    def push_many(self, iq):
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        assert iq.shape[0] % self.iq_len == 0
        v = iq.view(np.float32)
        took = 0
        for k in range(iq.shape[0] // self.iq_len):
            took += lib().orc_channel_push(C.byref(self.c), v[2 * k * self.iq_len: 2 * (k + 1) * self.iq_len])
        return took

    def push_stream(self, iq):
        """Whole iq_len blocks, then one trailing shorter block (a multiple of 4*D samples) pushed with
        iq_len temporarily set to its length -- how the GPU path accounts a non-multiple commit."""
        iq = np.ascontiguousarray(iq, dtype=np.complex64)
        whole = iq.shape[0] // self.iq_len * self.iq_len
        took = self.push_many(iq[:whole]) if whole else 0
        rest = iq.shape[0] - whole
        if rest:
            self.c.iq_len = rest
            took += lib().orc_channel_push(C.byref(self.c), np.ascontiguousarray(iq[whole:]).view(np.float32))
            self.c.iq_len = self.iq_len
        return took

    def boundary(self, epoch_s, want_f32=False):
        """-> None (frame discarded) or dict(i16, t_start, factor[, f32])."""
        i16 = np.empty(self.frame_len, np.int16)
        t0 = C.c_uint64(0); fac = C.c_float(0)
        f32 = np.empty(self.frame_len, np.float32) if want_f32 else None
        got = lib().orc_channel_boundary(C.byref(self.c), epoch_s, i16, C.byref(t0),
                                         f32.ctypes.data if want_f32 else None, C.byref(fac))
        if not got:
            return None
        r = dict(i16=i16, t_start=t0.value, factor=np.float32(fac.value))
        if want_f32:
            r["f32"] = f32
        return r

    @property
    def fill(self):
        return int(self.c.fill[self.c.wr])

    @property
    def dropped(self):
        return int(self.c.dropped_blocks)


def frame_len(mode):
    return int(lib().orc_frame_len(mode.encode()))


def prepare_audio(buf, mode, scale_ft=0.90, scale_wspr=0.20):
    """Returns (scaled copy, factor, peak)."""
    b = np.array(buf, dtype=np.float32, copy=True)
    pk = C.c_float(0)
    f = lib().orc_prepare_audio(b, b.shape[0], mode.encode(), scale_ft, scale_wspr, C.byref(pk))
    return b, np.float32(f), np.float32(pk.value)


def to_int16(buf):
    b = np.ascontiguousarray(buf, dtype=np.float32)
    o = np.empty(b.shape[0], np.int16)
    lib().orc_to_int16(b, b.shape[0], o)
    return o


def wav_header(n_samples):
    h = C.create_string_buffer(46)
    lib().orc_wav_header(n_samples, h)
    return h.raw


def synth_iq(seed, n, fs=192000, tones_hz=(), amp=2.0e4, first=0):
    """Portable synthetic IQ: integer noise (sigma~1182) + table-lookup tones.  complex64[n]."""
    x = np.empty(2 * n, np.float32)
    lib().orc_synth_noise(seed, first, n, x)
    if len(tones_hz):
        f = np.ascontiguousarray(tones_hz, dtype=np.float64)
        lib().orc_synth_add_tones(fs, first, n, f, len(f), amp, x)
    return x.view(np.complex64)


def checksum(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().orc_checksum_f32(x, x.shape[0]))


def crc32(a):
    a = np.ascontiguousarray(a)
    return int(lib().orc_crc32(a.ctypes.data, a.nbytes))


# ---- decoder hand-off formats (handoff_oracle.c; DecoderPool.hpp:58-171, 379-395, 451-590, 634-659, 1007-1046) ----
def decoder_block_bytes(js8=False):
    return int(lib().orc_js8_block_bytes() if js8 else lib().orc_jt9_block_bytes())


def decoder_block_offset(name, js8=False):
    """Byte offset of a member of the jt9/js8 shared-memory block, or -1."""
    return int((lib().orc_js8_offset if js8 else lib().orc_jt9_offset)(name.encode()))


def decoder_block(mode, audio_i16, decodedepth=3, highest_hz=3000, js8=False):
    """The filled shared-memory block as uint8[...]; None for a mode the jt9 route rejects."""
    audio = np.ascontiguousarray(audio_i16, dtype=np.int16)
    blk = np.empty(decoder_block_bytes(js8), np.uint8)
    if js8:
        rc = lib().orc_js8_fill(blk.ctypes.data, decodedepth, highest_hz, audio, audio.shape[0])
    else:
        rc = lib().orc_jt9_fill(blk.ctypes.data, mode.encode(), decodedepth, highest_hz, audio, audio.shape[0])
    return blk if rc == 0 else None


def decoder_route(mode, transfer_shmem=True):
    return "shmem" if lib().orc_decoder_route(mode.encode(), int(transfer_shmem)) else "wavefile"


def decoder_command(mode, target, shmem_route, numjt9threads=3, decodedepth=3, highest_decode_hz=3000, wspr_cycles=3000,
                    trperiod=0.0):
    app, opts = C.create_string_buffer(64), C.create_string_buffer(1024)
    rc = lib().orc_decoder_command(mode.encode(), int(shmem_route), numjt9threads, decodedepth, highest_decode_hz,
                                   wspr_cycles, float(trperiod), str(target).encode(), app, 64, opts, 1024)
    return (app.value.decode(), opts.value.decode()) if rc == 0 else None


# ---- FT4 coherent sync (ft4sync_oracle.c; PARITY UNPINNED: ft4_downsample / sync4d / ft4_decode search restated) ----
class _Ft4Sync(C.Structure):
    _fields_ = [("f0_hz", C.c_float), ("f1_hz", C.c_float), ("dt_s", C.c_float), ("sync", C.c_float),
                ("ibest", C.c_int32), ("idf", C.c_int32), ("seg", C.c_int32), ("cand", C.c_int32)]


def ft4_bigspec(frame_i16):
    """complex64[36289]: spectrum of the frame's first 72576 samples."""
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    assert fr.shape[0] >= 72576
    out = np.empty(2 * 36289, np.float32)
    lib().orc_ft4_bigspec(fr, out)
    return out.view(np.complex64)


def ft4_downsample(cx, f0_hz):
    """(complex64[4032] normalised baseband at 666.7 Hz, i0)."""
    cxf = np.ascontiguousarray(cx).view(np.float32)
    out = np.empty(2 * 4032, np.float32)
    i0 = lib().orc_ft4_downsample(cxf, float(f0_hz), out)
    return out.view(np.complex64), int(i0)


def ft4_sync4d(cd, i0, idf=0):
    return float(lib().orc_ft4_sync4d(np.ascontiguousarray(cd).view(np.float32), int(i0), int(idf)))


def ft4_search(cd, f0_hz, cand=0):
    """The three-segment coarse+fine search of ft4_decode: list of dicts (f0_hz, f1_hz, dt_s, sync, ibest, idf, seg, cand)."""
    buf = (_Ft4Sync * 3)()
    n = lib().orc_ft4_search(np.ascontiguousarray(cd).view(np.float32), float(f0_hz), int(cand), buf, 3)
    return [dict(f0_hz=b.f0_hz, f1_hz=b.f1_hz, dt_s=b.dt_s, sync=b.sync, ibest=b.ibest, idf=b.idf, seg=b.seg, cand=b.cand)
            for b in buf[:n]]


def ft4_sync_all(frame_i16, cands):
    """Refine every getcandidates4 candidate (tuples with freq_hz at index 3) of a frame."""
    cx = ft4_bigspec(frame_i16)
    out = []
    for k, c in enumerate(cands):
        cd, _ = ft4_downsample(cx, np.float32(c[3]))
        out += ft4_search(cd, np.float32(c[3]), k)
    return out


class _Spot(C.Structure):
    _fields_ = [("snr_db", C.c_int32), ("dt_s", C.c_float), ("freq_hz", C.c_uint32), ("has_locator", C.c_int32),
                ("call", C.c_char * 16), ("locator", C.c_char * 8), ("message", C.c_char * 64), ("drift", C.c_int32), ("dbm", C.c_int32)]


def parse_decode_line(mode, line, base_freq_hz):
    """spot_oracle.c: one jt9 stdout line (OutputHandler.cpp:505-621, 924-1128)."""
    sp = _Spot()
    rc = lib().orc_parse_decode_line(mode.encode(), line.encode(), int(base_freq_hz), C.byref(sp))
    return dict(status=("ok", "unhandled", "skip")[rc], snr_db=sp.snr_db, dt_s=sp.dt_s, freq_hz=sp.freq_hz,
                call=sp.call.decode(), locator=sp.locator.decode() if sp.has_locator else None, message=sp.message.decode(), drift=sp.drift, dbm=sp.dbm)


# ---- host-service rules (host_oracle.c; CWSL_DIGI.cpp:174-451, 857-887; CWSL_Utils.hpp:28-55) ----
def clock_sim(group, start_ms, end_ms, max_fires=4096):
    """Fire times (UTC ms) of the reference's polling thread of `group`, run in virtual time over [start, end)."""
    buf = np.zeros(max_fires, np.uint64)
    n = lib().orc_clock_sim(int(group), int(start_ms), int(end_ms), buf.ctypes.data, max_fires)
    return [int(x) for x in buf[:min(n, max_fires)]]


def pool_sizing(counts, decoderburden=1.0, n_decoders=None):
    arr = (C.c_int * 8)(*[int(x) for x in counts])
    nj, nw = C.c_int(), C.c_int()
    lib().orc_pool_sizing(arr, float(decoderburden), int(sum(counts) if n_decoders is None else n_decoders), C.byref(nj), C.byref(nw))
    return nj.value, nw.value


def find_band(bands, f_hz):
    lo = (C.c_int64 * len(bands))(*[int(b[0]) for b in bands])
    fs = (C.c_uint32 * len(bands))(*[int(b[1]) for b in bands])
    return int(lib().orc_find_band(lo, fs, len(bands), int(f_hz)))


def bench_cpu(threads, slots, fs=192000, iq_len=2048, n_per_slot=2880000):
    """Wall seconds for `threads` channels x `slots` FT8 slots on the host cores (reference shape)."""
    return float(lib().orc_bench_cpu(threads, slots, fs, iq_len, n_per_slot))


def bench_cpu_finalize(threads, reps):
    """Wall seconds of prepareAudio + int16 + the frame memset (Instance.cpp:294-338, 238-241, 213) for `threads` threads x `reps` FT8 frames."""
    L = lib()
    L.orc_bench_finalize.restype = C.c_double
    L.orc_bench_finalize.argtypes = [C.c_int, C.c_int]
    return float(L.orc_bench_finalize(int(threads), int(reps)))


class _Cand(C.Structure):
    _fields_ = [("freq_bin", C.c_int32), ("time_step", C.c_int32), ("sync", C.c_float),
                ("freq_hz", C.c_float), ("dt_s", C.c_float)]


def ft8_spectra(frame_i16, nbins):
    """PARITY UNPINNED (sync_oracle.h).  float32[372, nbins] symbol power spectra of an int16 frame."""
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    assert fr.shape[0] >= 180000
    out = np.empty(372 * nbins, np.float32)
    assert lib().orc_ft8_spectra(fr, out, nbins) == 0
    return out.reshape(372, nbins)


ORDER = {"sync": 0, "freq": 1, 0: 0, 1: 1}          # cwslg_set_candidate_order: strongest first (default) / ascending frequency, cut in that order


def ft8_tdiff_close(lag_i, lag_j):
    """sync8's `tdiff < 0.04` in the float32 expression the restatement and the kernel use."""
    return bool(lib().orc_ft8_tdiff_close(int(lag_i), int(lag_j)))


def ft8_sync(frame_i16, f_lo_hz=200, f_hi_hz=3000, syncmin=1.5, maxcand=200, want_arrays=False, order="sync"):
    """PARITY UNPINNED.  -> list of (freq_bin, time_step, sync, freq_hz, dt_s) [, dict of red/jpeak arrays]."""
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    assert fr.shape[0] >= 180000
    buf = (_Cand * maxcand)()
    red = np.zeros(1921, np.float32); red2 = np.zeros(1921, np.float32)
    jp = np.zeros(1921, np.int32); jp2 = np.zeros(1921, np.int32)
    n = lib().orc_ft8_sync_ordered(fr, f_lo_hz, f_hi_hz, syncmin, maxcand, ORDER[order], C.addressof(buf), maxcand,
                                   red.ctypes.data, jp.ctypes.data, red2.ctypes.data, jp2.ctypes.data)
    assert n >= 0
    cands = [(buf[k].freq_bin, buf[k].time_step, buf[k].sync, buf[k].freq_hz, buf[k].dt_s) for k in range(n)]
    if want_arrays:
        return cands, dict(red=red, red2=red2, jpeak=jp, jpeak2=jp2)
    return cands


def ft4_spectra(frame_i16):
    """PARITY UNPINNED.  float32[122, 1153] Nuttall-windowed symbol power spectra of an FT4 int16 frame."""
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    assert fr.shape[0] >= 72576
    out = np.empty(122 * 1153, np.float32)
    assert lib().orc_ft4_spectra(fr, out) == 0
    return out.reshape(122, 1153)


def ft4_candidates(frame_i16, fa_hz=200.0, fb_hz=4000.0, syncmin=1.2, maxcand=200, want_arrays=False, order="sync"):
    """PARITY UNPINNED.  -> list of (freq_bin, 0, height, freq_hz, 0.0) [, dict(savsm, sbase)]."""
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    assert fr.shape[0] >= 72576
    buf = (_Cand * maxcand)()
    sm = np.zeros(1153, np.float32); sb = np.zeros(1153, np.float32)
    n = lib().orc_ft4_candidates_ordered(fr, fa_hz, fb_hz, syncmin, maxcand, ORDER[order], C.addressof(buf), maxcand, sm.ctypes.data, sb.ctypes.data)
    assert n >= 0
    cands = [(buf[k].freq_bin, buf[k].time_step, buf[k].sync, buf[k].freq_hz, buf[k].dt_s) for k in range(n)]
    return (cands, dict(savsm=sm, sbase=sb)) if want_arrays else cands


def bench_cpu_reference(threads, slots, fs=192000, iq_len=2048, n_per_slot=2880000):
    """Wall seconds of the REFERENCE's own SSBD<float> loop (oracle/_ref) for `threads` channels x `slots` slots."""
    return float(ref().ref_bench_cpu(threads, slots, fs, iq_len, n_per_slot))


# ---------------------------------------------------------------------------------------------- 120 s modes (row a14)
class _WsprCand(C.Structure):
    _fields_ = [("freq_hz", C.c_float), ("snr_db", C.c_float), ("drift", C.c_float), ("sync", C.c_float), ("shift", C.c_int32)]


class _Fst4wCand(C.Structure):
    _fields_ = [("freq_hz", C.c_float), ("snr", C.c_float), ("bin", C.c_int32), ("pad_", C.c_int32)]


def fftb(na, nb, x, inverse=False):
    """spec B transform of na*nb complex points (longsync_oracle.c)."""
    re = np.ascontiguousarray(x.real, dtype=np.float32).copy()
    im = np.ascontiguousarray(x.imag, dtype=np.float32).copy()
    L = lib()
    L.orc_fftb.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    assert L.orc_fftb(na, nb, re.ctypes.data, im.ctypes.data, 1 if inverse else 0) == 0
    return re + 1j * im


def wspr_downsample(frame_i16):
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    i = np.empty(46080, np.float32); q = np.empty(46080, np.float32)
    L = lib()
    L.orc_wspr_downsample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    assert L.orc_wspr_downsample(fr.ctypes.data, len(fr), i.ctypes.data, q.ctypes.data) == 0
    return i, q


def wspr_search(frame_i16, want_arrays=False):
    """-> list of (freq_hz, snr_db, drift, sync, shift) [, dict(idat, qdat, ps[512,359], smspec[411])]."""
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    out = (_WsprCand * 200)()
    arr = dict(idat=np.empty(46080, np.float32), qdat=np.empty(46080, np.float32), ps=np.empty((512, 359), np.float32),
               smspec=np.empty(411, np.float32))
    L = lib()
    L.orc_wspr_search.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    n = L.orc_wspr_search(fr.ctypes.data, len(fr), out, 200, arr["idat"].ctypes.data, arr["qdat"].ctypes.data,
                          arr["ps"].ctypes.data, arr["smspec"].ctypes.data)
    assert n >= 0
    cands = [(c.freq_hz, c.snr_db, c.drift, c.sync, c.shift) for c in out[:n]]
    return (cands, arr) if want_arrays else cands


def fst4w_candidates(frame_i16, nfa_hz=1400, nfb_hz=1600, minsync=1.2, want_arrays=False):
    """-> list of (freq_hz, snr, bin) [, dict(s2[n], band[nband])]."""
    fr = np.ascontiguousarray(frame_i16, dtype=np.int16)
    out = (_Fst4wCand * 100)()
    s2 = np.zeros(8192, np.float32); band = np.zeros(32000, np.float32)
    L = lib()
    L.orc_fst4w_candidates.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    n = L.orc_fst4w_candidates(fr.ctypes.data, len(fr), nfa_hz, nfb_hz, minsync, out, 100, s2.ctypes.data, len(s2), band.ctypes.data)
    assert n >= 0
    cands = [(c.freq_hz, c.snr, c.bin) for c in out[:n]]
    return (cands, dict(s2=s2, band=band)) if want_arrays else cands

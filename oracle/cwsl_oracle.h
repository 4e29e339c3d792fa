/*
 * cwsl_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99) of the one hot path of CWSL_DIGI that this
 * repository re-implements for MI355X: the per-Instance DSP chain
 *   SSBD (NCO mix + windowed-sinc low-pass + decimate + Fs/4 up-shift)
 *   -> slot framing -> peak normalise -> int16,
 * plus the 12 kHz WAV container.  Every function cites the reference
 * file:line it follows (paths relative to /root/reference/source/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * include, link or load this.  The product library (libcwslgpu.so) never
 * does; it fails loudly without a GPU instead of falling back here.
 *
 * Parity pinning: the reference ships no tests or golden vectors
 * (SURVEY.md section 4).  This restatement is pinned against the reference
 * itself: oracle/_ref/libcwsl_ref.so is the unmodified SSBD.hpp/LowPass.hpp
 * compiled in place, and tests/test_oracle_vs_ref.py + the committed
 * fixtures in tests/golden/ (made by tests/gen_golden.py from that build)
 * require bit-identical taps, tone, phasor trace and audio.  The slot framing (orc_channel_*) is pinned the same
 * way against Instance.cpp's call sequences replayed on the reference's own ring_buffer_t / sample_buffer_t
 * containers (ring_buffer.h, decode_audio_buffer.h compile as they are): same drop decisions, epochs, sample
 * counts and frames (tests/test_oracle_vs_ref.py::test_framing_matches_reference_containers).
 *
 * Build of record: gcc -std=c99 -O2 -ffp-contract=off (no -march=native,
 * no -ffast-math) -- the same floating-point contract as the reference
 * build used for the fixtures.
 */
#ifndef CWSL_ORACLE_H
#define CWSL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK                 0
#define ORC_ERR_RATIO         -1   /* SSBD.hpp:54-55  "Fs/B must be an even integer >= 4" */
#define ORC_ERR_BAND_LOW      -2   /* SSBD.hpp:100-101 "Signal outside of band (low)"     */
#define ORC_ERR_BAND_HIGH     -3   /* SSBD.hpp:102-103 "Signal outside of band (high)"    */
#define ORC_ERR_ALLOC         -4
#define ORC_ERR_MODE          -5   /* CWSL_DIGI.hpp:110-112 "Unhandled mode"              */

#define ORC_NUM_WS   32            /* NumWS = FiltOrder/BlockSize = 2*latency*2 = 32 for latency_log2=3 */
#define ORC_WAVE_SR  12000u        /* CWSL_DIGI.hpp:51 */
#define ORC_SSB_BW   6000u         /* CWSL_DIGI.hpp:52 */

/* ---- demodulator state (one per channel per slot) ---- */
typedef struct {
    uint64_t fs, bw;
    int      usb;
    float    sign;          /* +1 USB, -1 LSB (SSBD.hpp:110) */
    uint32_t block;         /* BlockSize = Fs/B/2 (SSBD.hpp:71) */
    uint32_t ntaps;         /* FiltOrder = 8*2*Fs/B (SSBD.hpp:62) */
    float   *taps;          /* normalised windowed sinc (SSBD.hpp:63-68) */
    float   *tone_re, *tone_im;    /* block-local mixing tone (SSBD.hpp:112-113) */
    float    inc_re, inc_im;       /* per-block phasor step (SSBD.hpp:114) */
    float    ph_re, ph_im;         /* running phasor (SSBD.hpp:121,174) */
    float    ws_re[ORC_NUM_WS], ws_im[ORC_NUM_WS];  /* overlap-add slots (SSBD.hpp:75-77) */
    uint32_t head;          /* index of next output slot (SSBD.hpp:177-179) */
    float    phase_delta;   /* kept for fixtures */
} orc_demod_t;

int  orc_lowpass_design(size_t order, double bandwidth, float *taps);           /* LowPass.hpp:16-35 */
int  orc_demod_open(orc_demod_t *d, uint64_t fs, uint64_t bw, double f_hz, int usb); /* SSBD.hpp:48-83,97-123 */
void orc_demod_close(orc_demod_t *d);
/* SSBD::Tune(F, isUSB, reset = true) on an open demodulator (SSBD.hpp:96-123) */
int  orc_demod_tune(orc_demod_t *d, double f_hz, int usb);
/* SSBD::Tune(F, isUSB, reset): reset = 0 keeps workspace, index and phase (SSBD.hpp:116-121 skipped) */
int  orc_demod_tune_ex(orc_demod_t *d, double f_hz, int usb, int reset);
/* one Iterate(): consumes 4*block complex samples, emits 4 floats (SSBD.hpp:127-137,160-183) */
void orc_demod_iterate(orc_demod_t *d, const float *iq_ri, float *out4);
/* n_complex must be a multiple of 4*block; phase_trace (optional) gets the phasor before every block */
void orc_demod_run(orc_demod_t *d, const float *iq_ri, uint64_t n_complex, float *out, float *phase_trace);

/* ---- slot framing (Instance.cpp) ---- */
double orc_rx_period(const char *mode);                    /* CWSL_DIGI.hpp:64-113 ; <0 if unknown */
size_t orc_frame_len(const char *mode);                    /* Instance.cpp:149 : 12000*(period+5) */
/* Instance.cpp:294-338 ; scales buf in place, returns the factor; *peak_out = maxVal used */
float  orc_prepare_audio(float *buf, size_t n, const char *mode,
                         float scale_ft, float scale_wspr, float *peak_out);
void   orc_to_int16(const float *buf, size_t n, int16_t *out);   /* Instance.cpp:238-241 */

typedef struct {
    char     mode[16];
    uint64_t fs;
    uint32_t iq_len;            /* Receiver block length (Receiver.hpp:88) */
    int32_t  demod_hz;          /* calibratedSSBFreq - LO (Instance.cpp:183) */
    float    scale_ft, scale_wspr;
    size_t   frame_len;
    float   *frame[2];          /* af_buffer.recs[0..1].buf (Instance.hpp:95) */
    uint64_t fill[2];           /* recs[k].write_index */
    uint64_t t0[2];             /* recs[k].startEpochTime */
    uint32_t wr, rd;            /* ring write/read index (ring_buffer.h:34-35) */
    orc_demod_t demod;
    uint64_t dropped_blocks;    /* "af buffer full" events (Instance.cpp:268-271) */
} orc_channel_t;

int  orc_channel_open(orc_channel_t *c, const char *mode, uint64_t fs, uint32_t iq_len,
                      int32_t demod_hz, float scale_ft, float scale_wspr);   /* Instance.cpp:121-176,187 */
void orc_channel_close(orc_channel_t *c);
/* one IQ block of iq_len complex samples (Instance.cpp:260-276); returns 1 if consumed, 0 if dropped */
int  orc_channel_push(orc_channel_t *c, const float *iq_ri);
/* slot boundary (Instance.cpp:203-253).  Returns 1 and fills out_i16[frame_len], *t_start when a
 * frame is emitted, 0 when the finished frame is discarded (startEpochTime == 0).
 * audio_f32 (optional, frame_len floats) receives the frame BEFORE prepareAudio scaling. */
int  orc_channel_boundary(orc_channel_t *c, uint64_t epoch_s, int16_t *out_i16,
                          uint64_t *t_start, float *audio_f32, float *factor_out);

/* ---- WAV container (WaveFile.hpp:19-35,87-135): 46-byte header ---- */
#define ORC_WAV_HDR_BYTES 46
void orc_wav_header(uint32_t n_samples, uint8_t hdr[ORC_WAV_HDR_BYTES]);
int  orc_wav_write(const char *path, const int16_t *pcm, uint32_t n_samples);

/* ---- portable synthetic IQ (builder-defined) ---- */
uint64_t orc_mix64(uint64_t z);
/* noise: integer-valued/32 floats, sigma ~1182, exact in any IEEE-754 implementation */
void orc_synth_noise(uint64_t seed, uint64_t first_sample, uint64_t n_complex, float *iq_ri);
/* add K complex tones; tone k has amplitude amp, frequency f_hz[k] (Hz, relative to LO);
 * phase is an exact uint32 accumulator looked up in a 4096-entry table made by sin()/cos() */
void orc_synth_add_tones(uint64_t fs, uint64_t first_sample, uint64_t n_complex,
                         const double *f_hz, int k_tones, float amp, float *iq_ri);

/* CPU baseline ("port"): `threads` channels in parallel (one pthread each, the reference's
 * thread-per-Instance shape), `slots` FT8 slots of n_per_slot samples each; returns wall seconds. */
double orc_bench_cpu(int threads, int slots, uint64_t fs, uint32_t iq_len, uint64_t n_per_slot);
double orc_bench_finalize(int threads, int reps);

/* position-weighted checksum used by fixtures: sum_k (1+(k%251)) * x[k] in double */
double orc_checksum_f32(const float *x, size_t n);
uint32_t orc_crc32(const void *data, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif

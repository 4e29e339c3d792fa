/*
 * cwsl_oracle.c -- TEST INFRASTRUCTURE ONLY (see cwsl_oracle.h).
 *
 * Plain-C CPU restatement of the CWSL_DIGI per-channel DSP chain.  It keeps
 * the reference's *shape* (recursive overlap-add block filter, recursive
 * float32 phasor, one state object per channel) so that it can double as the
 * "port" CPU baseline timed by bench.py, and it keeps the reference's exact
 * float operation order so that it is bit-identical to oracle/_ref.
 *
 * Floating-point contract: compile with -O2 -ffp-contract=off.  All complex
 * products are written out as four multiplies, one subtract, one add -- the
 * sequence g++ emits for std::complex<float> operator* without fast-math.
 */
#include "cwsl_oracle.h"

#include <complex.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* LowPass.hpp:13 */
static const double kPi = 3.14159265358979323846;

/* ------------------------------------------------------------------ */
/* LowPass.hpp:16-35 -- Hamming-weighted sinc, `order` taps, tap 0 = 0, */
/* centre tap = 1, symmetric; evaluated in double, stored as float.     */
int orc_lowpass_design(size_t order, double bandwidth, float *taps)
{
    if (!taps || order < 2) return ORC_ERR_ALLOC;
    const size_t half = order / 2;
    taps[0] = 0.0f;
    taps[half] = 1.0f;
    const double x0 = -1.0 * (double)order / 2;                 /* :26 */
    for (size_t n = 1; n < half; ++n) {                         /* :27 */
        const double xpi = (x0 + (double)n) * kPi * bandwidth;  /* :28 */
        const double w = 0.54 - 0.46 * cos(2.0 * kPi * (double)n / (double)order);
        const double y = sin(xpi) / xpi * w;                    /* :29 */
        taps[n] = (float)y;
        taps[order - n] = (float)y;
    }
    return ORC_OK;
}

/* complex<float>(re, im) without arithmetic, so that a -0.0f imaginary part survives
 * (phase_delta*0 is -0.0f for positive tuning offsets and cexpf keeps that sign). */
static float complex make_cf(float re, float im)
{
    float complex z;
    __real__ z = re;
    __imag__ z = im;
    return z;
}

/* ------------------------------------------------------------------ */
/* SSBD.hpp:48-83 (constructor) + :97-123 (Tune, reset=true)           */
int orc_demod_open(orc_demod_t *d, uint64_t fs, uint64_t bw, double f_hz, int usb)
{
    memset(d, 0, sizeof(*d));
    /* :54 -- integer arithmetic exactly as written there */
    if (bw == 0 || (fs / bw / 2) * 2 * bw != fs || fs < 4 * bw) return ORC_ERR_RATIO;
    const uint64_t latency = 1u << 3;                           /* latency_log2 = 3 default (:49) */
    d->fs = fs; d->bw = bw; d->usb = usb ? 1 : 0;
    d->ntaps = (uint32_t)(latency * 2 * fs / bw);               /* :62 */
    d->block = (uint32_t)(fs / bw / 2);                         /* :71 */
    if (d->ntaps / d->block != ORC_NUM_WS) return ORC_ERR_RATIO;

    d->taps = (float *)malloc(sizeof(float) * d->ntaps);
    d->tone_re = (float *)malloc(sizeof(float) * d->block);
    d->tone_im = (float *)malloc(sizeof(float) * d->block);
    if (!d->taps || !d->tone_re || !d->tone_im) { orc_demod_close(d); return ORC_ERR_ALLOC; }

    orc_lowpass_design(d->ntaps, (double)bw / (double)fs, d->taps);   /* :63 */
    float acc = 0.0f;                                           /* :66-68, float running sum */
    for (uint32_t n = 0; n < d->ntaps; ++n) acc += d->taps[n];
    for (uint32_t n = 0; n < d->ntaps; ++n) d->taps[n] /= acc;

    {
        const int rc = orc_demod_tune(d, f_hz, usb);            /* :81 the constructor ends in Tune(F, isUSB) */
        if (rc != ORC_OK) { orc_demod_close(d); return rc; }
    }
    return ORC_OK;
}

/* SSBD::Tune(F, isUSB, reset = true), SSBD.hpp:96-123.  On a range error the object is left untouched (the throw
 * happens before anything is stored). */
int orc_demod_tune(orc_demod_t *d, double f_hz, int usb) { return orc_demod_tune_ex(d, f_hz, usb, 1); }

/* The same with Tune's third argument: reset = 0 keeps workspace, index and phase (:116-121 skipped), so the partial sums of the
 * last 31 blocks -- mixed with the OLD tone and phase -- stay in the workspace and the phasor continues from its current value
 * with the new step. */
int orc_demod_tune_ex(orc_demod_t *d, double f_hz, int usb, int reset)
{
    const uint64_t fs = d->fs, bw = d->bw;
    /* range checks use integer Fs/2 promoted to double (:100-103) */
    const double half_band = (double)(fs / 2);
    if (fabs(f_hz) > half_band) return ORC_ERR_BAND_LOW;
    if (fabs(f_hz + (double)bw * (usb ? 1.0 : -1.0)) > half_band) return ORC_ERR_BAND_HIGH;
    d->usb = usb ? 1 : 0;

    d->sign = usb ? 1.0f : -1.0f;                               /* :110 */
    /* :111  -2.0*PI*(F + sign*B/2.0)/Fs : sign*B is a float product, the rest double */
    const float sb = d->sign * (float)bw;
    const double pd = -2.0 * kPi * (f_hz + (double)sb / 2.0) / (double)fs;
    d->phase_delta = (float)pd;
    for (uint32_t n = 0; n < d->block; ++n) {                   /* :112-113 */
        const float complex e = cexpf(make_cf(0.0f, d->phase_delta * (float)n));
        d->tone_re[n] = crealf(e);
        d->tone_im[n] = cimagf(e);
    }
    {                                                           /* :114 */
        const float complex e = cexpf(make_cf(0.0f, d->phase_delta * (float)d->block));
        d->inc_re = crealf(e);
        d->inc_im = cimagf(e);
    }
    if (reset) {                                                /* :116-121 */
        for (int k = 0; k < ORC_NUM_WS; ++k) { d->ws_re[k] = 0.0f; d->ws_im[k] = 0.0f; }
        d->head = 0;
        d->ph_re = 1.0f; d->ph_im = 0.0f;
    }
    return ORC_OK;
}

void orc_demod_close(orc_demod_t *d)
{
    free(d->taps); free(d->tone_re); free(d->tone_im);
    d->taps = d->tone_re = d->tone_im = NULL;
}

/* SSBD.hpp:160-183 -- one block in, one complex sample out */
static void demod_block(orc_demod_t *d, const float *in_ri, float *z_re, float *z_im)
{
    const uint32_t nb = d->block;
    const float *h = d->taps;
    for (uint32_t n = 0; n < ORC_NUM_WS; ++n) {                 /* :164 */
        float sr = 0.0f, si = 0.0f;
        for (uint32_t m = 0; m < nb; ++m) {                     /* :166-169 */
            const float xr = in_ri[2 * m], xi = in_ri[2 * m + 1];
            const float tr = d->tone_re[m], ti = d->tone_im[m];
            const float mr = xr * tr - xi * ti;                 /* in[m]*tone[m] */
            const float mi = xr * ti + xi * tr;
            const float c = h[m + n * nb];
            sr += mr * c;                                       /* (..)*filter[], then += */
            si += mi * c;
        }
        const float pr = sr * d->ph_re - si * d->ph_im;         /* sum*phase (:170) */
        const float pi = sr * d->ph_im + si * d->ph_re;
        const uint32_t slot = (ORC_NUM_WS - n - 1 + d->head) & (ORC_NUM_WS - 1);
        d->ws_re[slot] += pr;
        d->ws_im[slot] += pi;
    }
    {                                                           /* :174 phase *= phase_inc */
        const float nr = d->ph_re * d->inc_re - d->ph_im * d->inc_im;
        const float ni = d->ph_re * d->inc_im + d->ph_im * d->inc_re;
        d->ph_re = nr; d->ph_im = ni;
    }
    *z_re = d->ws_re[d->head];                                  /* :177-179 */
    *z_im = d->ws_im[d->head];
    d->ws_re[d->head] = 0.0f;
    d->ws_im[d->head] = 0.0f;
    d->head = (d->head + 1) & (ORC_NUM_WS - 1);
}

/* SSBD.hpp:127-137 */
void orc_demod_iterate(orc_demod_t *d, const float *iq_ri, float *out4)
{
    float zr, zi;
    const uint32_t nb = d->block;
    demod_block(d, iq_ri + 0 * 2 * nb, &zr, &zi); out4[0] = +zr;
    demod_block(d, iq_ri + 1 * 2 * nb, &zr, &zi); out4[1] = -zi * d->sign;
    demod_block(d, iq_ri + 2 * 2 * nb, &zr, &zi); out4[2] = -zr;
    demod_block(d, iq_ri + 3 * 2 * nb, &zr, &zi); out4[3] = +zi * d->sign;
}

void orc_demod_run(orc_demod_t *d, const float *iq_ri, uint64_t n_complex, float *out, float *phase_trace)
{
    const uint64_t step = 4ull * d->block;
    for (uint64_t n = 0; n < n_complex; n += step) {
        if (phase_trace) {
            float pr = d->ph_re, pi = d->ph_im;
            for (int k = 0; k < 4; ++k) {
                const uint64_t b = n / d->block + (uint64_t)k;
                phase_trace[2 * b] = pr; phase_trace[2 * b + 1] = pi;
                const float nr = pr * d->inc_re - pi * d->inc_im;
                const float ni = pr * d->inc_im + pi * d->inc_re;
                pr = nr; pi = ni;
            }
        }
        orc_demod_iterate(d, iq_ri + 2 * n, out + n / d->block);
    }
}

/* ------------------------------------------------------------------ */
/* CWSL_DIGI.hpp:64-113 */
double orc_rx_period(const char *mode)
{
    static const struct { const char *m; double p; } tab[] = {
        {"FT8", 15.0}, {"JS8", 15.0}, {"FT4", 7.5}, {"WSPR", 120.0}, {"Q65-30", 30.0},
        {"JT65", 60.0}, {"FST4-60", 60.0}, {"FST4-120", 120.0}, {"FST4-300", 300.0},
        {"FST4-900", 900.0}, {"FST4-1800", 1800.0}, {"FST4W-120", 120.0},
        {"FST4W-300", 300.0}, {"FST4W-900", 900.0}, {"FST4W-1800", 1800.0},
    };
    for (size_t k = 0; k < sizeof(tab) / sizeof(tab[0]); ++k)
        if (strcmp(mode, tab[k].m) == 0) return tab[k].p;
    return -1.0;
}

/* Instance.cpp:149 : (size_t)((double)SSB_SR * (double)(getRXPeriod(mode)+5)) ; period is float */
size_t orc_frame_len(const char *mode)
{
    const double p = orc_rx_period(mode);
    if (p < 0) return 0;
    const float pf = (float)p + 5;
    return (size_t)((double)ORC_WAVE_SR * (double)pf);
}

/* Instance.cpp:294-338 */
float orc_prepare_audio(float *buf, size_t n, const char *mode,
                        float scale_ft, float scale_wspr, float *peak_out)
{
    float hi = -3.402823466e+38f;                               /* numeric_limits<float>::lowest() */
    for (size_t k = 0; k < n; ++k) if (buf[k] > hi) hi = buf[k];
    float lo = 3.402823466e+38f;
    for (size_t k = 0; k < n; ++k) if (buf[k] < lo) lo = buf[k];
    if (fabsf(lo) > hi) hi = fabsf(lo);                         /* :310-312 */
    const float clip = 32767.0f;                                /* pow(2.f,15.f)-1.f, CWSL_DIGI.hpp:55 */
    float factor = clip / (hi + 1.0f);                          /* :316 */
    if (strcmp(mode, "WSPR") == 0) factor *= scale_wspr;        /* :320-324, exact string match */
    else factor *= scale_ft;                                    /* :325-329 */
    for (size_t k = 0; k < n; ++k) buf[k] *= factor;            /* :332-334 */
    if (peak_out) *peak_out = hi;
    return factor;
}

/* Instance.cpp:238-241 : add 0.5f, then C truncation toward zero */
void orc_to_int16(const float *buf, size_t n, int16_t *out)
{
    for (size_t k = 0; k < n; ++k) out[k] = (int16_t)(buf[k] + 0.5f);
}

/* ------------------------------------------------------------------ */
/* Instance.cpp:121-176 (init) + :181-196 (first SSBD) */
int orc_channel_open(orc_channel_t *c, const char *mode, uint64_t fs, uint32_t iq_len,
                     int32_t demod_hz, float scale_ft, float scale_wspr)
{
    memset(c, 0, sizeof(*c));
    if (orc_rx_period(mode) < 0) return ORC_ERR_MODE;
    snprintf(c->mode, sizeof(c->mode), "%s", mode);
    c->fs = fs; c->iq_len = iq_len; c->demod_hz = demod_hz;
    c->scale_ft = scale_ft; c->scale_wspr = scale_wspr;
    c->frame_len = orc_frame_len(mode);
    for (int k = 0; k < 2; ++k) {
        c->frame[k] = (float *)calloc(c->frame_len, sizeof(float));    /* :154-156 */
        if (!c->frame[k]) { orc_channel_close(c); return ORC_ERR_ALLOC; }
    }
    /* Instance.cpp:187 : F is static_cast<float>(int32 demodFreq) widened to double */
    int rc = orc_demod_open(&c->demod, fs, ORC_SSB_BW, (double)(float)demod_hz, 1);
    if (rc != ORC_OK) { orc_channel_close(c); return rc; }
    return ORC_OK;
}

void orc_channel_close(orc_channel_t *c)
{
    free(c->frame[0]); free(c->frame[1]);
    c->frame[0] = c->frame[1] = NULL;
    orc_demod_close(&c->demod);
}

/* Instance.cpp:260-276 */
int orc_channel_push(orc_channel_t *c, const float *iq_ri)
{
    const uint32_t w = c->wr;
    /* :268 -- note: compares audio fill + *IQ* block length against size-1 */
    if (c->fill[w] + c->iq_len > c->frame_len - 1) { c->dropped_blocks++; return 0; }
    const uint32_t in_size = 4u * c->demod.block;               /* GetInSize() */
    const uint32_t dec = (uint32_t)(c->fs / ORC_WAVE_SR);       /* decRatio (:192) */
    float *dst = c->frame[w] + c->fill[w];                      /* :272 */
    for (uint32_t n = 0; n < c->iq_len; n += in_size)           /* :273-275 */
        orc_demod_iterate(&c->demod, iq_ri + 2 * n, dst + n / dec);
    c->fill[w] += c->iq_len / dec;                              /* :276 */
    return 1;
}

/* Instance.cpp:203-253 with the 2-deep ring of ring_buffer.h:92-128 */
int orc_channel_boundary(orc_channel_t *c, uint64_t epoch_s, int16_t *out_i16,
                         uint64_t *t_start, float *audio_f32, float *factor_out)
{
    const uint32_t nxt = (c->wr == 1) ? 0 : c->wr + 1;          /* get_next_write_index, size 2 */
    memset(c->frame[nxt], 0, c->frame_len * sizeof(float));     /* :213 */
    c->fill[nxt] = 0;                                           /* :214 */
    c->t0[nxt] = epoch_s;                                       /* :215 */
    c->wr = nxt;                                                /* :217 inc_write_index */
    const uint32_t cur = c->rd;                                 /* :221 pop_ref */
    c->rd = (c->rd == 1) ? 0 : c->rd + 1;
    const uint64_t started = c->t0[cur];
    if (started == 0) return 0;                                 /* :224-227 -- NOTE: no SSBD reset on this path */

    if (audio_f32) memcpy(audio_f32, c->frame[cur], c->frame_len * sizeof(float));
    float f = orc_prepare_audio(c->frame[cur], c->frame_len, c->mode,
                                c->scale_ft, c->scale_wspr, NULL);      /* :230 */
    if (factor_out) *factor_out = f;
    orc_to_int16(c->frame[cur], c->frame_len, out_i16);         /* :238-241 */
    if (t_start) *t_start = started;
    /* :251 -- brand-new SSBD: taps rebuilt, workspace zero, phasor (1,0) */
    orc_demod_close(&c->demod);
    orc_demod_open(&c->demod, c->fs, ORC_SSB_BW, (double)(float)c->demod_hz, 1);
    return 1;
}

/* ------------------------------------------------------------------ */
/* WaveFile.hpp:19-35 + :96-113.  WAVEFORMATEX is 18 bytes, header 46.  */
static void put_u32(uint8_t *p, uint32_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; p[2] = (v >> 16) & 255; p[3] = (v >> 24) & 255; }
static void put_u16(uint8_t *p, uint16_t v) { p[0] = v & 255; p[1] = (v >> 8) & 255; }

void orc_wav_header(uint32_t n_samples, uint8_t hdr[ORC_WAV_HDR_BYTES])
{
    const uint32_t data_len = n_samples * 2u;
    memcpy(hdr + 0, "RIFF", 4);
    put_u32(hdr + 4, ORC_WAV_HDR_BYTES + data_len - 8);         /* :99 */
    memcpy(hdr + 8, "WAVE", 4);
    memcpy(hdr + 12, "fmt ", 4);
    put_u32(hdr + 16, 18);                                      /* sizeof(WAVEFORMATEX) :103 */
    put_u16(hdr + 20, 1);                                       /* WAVE_FORMAT_PCM */
    put_u16(hdr + 22, 1);                                       /* nChannels */
    put_u32(hdr + 24, 12000);                                   /* nSamplesPerSec */
    put_u32(hdr + 28, 24000);                                   /* nAvgBytesPerSec */
    put_u16(hdr + 32, 2);                                       /* nBlockAlign */
    put_u16(hdr + 34, 16);                                      /* wBitsPerSample */
    put_u16(hdr + 36, 0);                                       /* cbSize */
    memcpy(hdr + 38, "data", 4);
    put_u32(hdr + 42, data_len);                                /* :113 */
}

int orc_wav_write(const char *path, const int16_t *pcm, uint32_t n_samples)
{
    uint8_t hdr[ORC_WAV_HDR_BYTES];
    orc_wav_header(n_samples, hdr);
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    int ok = fwrite(hdr, 1, sizeof(hdr), f) == sizeof(hdr) &&
             fwrite(pcm, 2, n_samples, f) == n_samples;
    fclose(f);
    return ok ? 0 : -1;
}

/* ------------------------------------------------------------------ */
/* Portable synthetic IQ.  Builder-defined; no reference counterpart.   */
uint64_t orc_mix64(uint64_t z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static float noise_from_bits(uint64_t r)
{
    /* Irwin-Hall(4) of 16-bit fields, centred, /32: every value is an exact float */
    const int32_t s = (int32_t)(r & 0xFFFF) + (int32_t)((r >> 16) & 0xFFFF) +
                      (int32_t)((r >> 32) & 0xFFFF) + (int32_t)((r >> 48) & 0xFFFF) - 131070;
    return (float)s * 0.03125f;
}

void orc_synth_noise(uint64_t seed, uint64_t first_sample, uint64_t n_complex, float *iq_ri)
{
    for (uint64_t k = 0; k < n_complex; ++k) {
        const uint64_t i = first_sample + k;
        iq_ri[2 * k]     = noise_from_bits(orc_mix64(seed ^ (2 * i) * 0xD1342543DE82EF95ull));
        iq_ri[2 * k + 1] = noise_from_bits(orc_mix64(seed ^ (2 * i + 1) * 0xD1342543DE82EF95ull));
    }
}

void orc_synth_add_tones(uint64_t fs, uint64_t first_sample, uint64_t n_complex,
                         const double *f_hz, int k_tones, float amp, float *iq_ri)
{
    static float tab_c[4096], tab_s[4096];
    static int ready = 0;
    if (!ready) {
        for (int j = 0; j < 4096; ++j) {
            tab_c[j] = (float)cos(2.0 * kPi * j / 4096.0);
            tab_s[j] = (float)sin(2.0 * kPi * j / 4096.0);
        }
        ready = 1;
    }
    for (int t = 0; t < k_tones; ++t) {
        const double cyc = f_hz[t] / (double)fs;                /* cycles/sample, |cyc| <= 0.5 */
        const uint32_t step = (uint32_t)(int64_t)llround(cyc * 4294967296.0);
        for (uint64_t k = 0; k < n_complex; ++k) {
            const uint32_t ph = (uint32_t)((first_sample + k) * (uint64_t)step);
            const uint32_t j = ph >> 20;
            iq_ri[2 * k]     += amp * tab_c[j];
            iq_ri[2 * k + 1] += amp * tab_s[j];
        }
    }
}

double orc_checksum_f32(const float *x, size_t n)
{
    double s = 0.0;
    for (size_t k = 0; k < n; ++k) s += (double)(1 + (k % 251)) * (double)x[k];
    return s;
}

uint32_t orc_crc32(const void *data, size_t nbytes)
{
    const uint8_t *p = (const uint8_t *)data;
    uint32_t c = 0xFFFFFFFFu;
    for (size_t k = 0; k < nbytes; ++k) {
        c ^= p[k];
        for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return c ^ 0xFFFFFFFFu;
}

/* ------------------------------------------------------------------ */
/* CPU baseline driver for bench.py ("port" kind): the reference's shape -- one thread per channel,
 * each running the recursive block filter over whole slots, then prepareAudio + int16.             */
#include <pthread.h>
#include <time.h>

typedef struct {
    uint64_t fs; uint32_t iq_len; uint64_t n_per_slot; int slots; int32_t demod_hz; uint64_t seed;
    double checksum; int rc;
} bench_job_t;

static void *bench_worker(void *arg)
{
    bench_job_t *j = (bench_job_t *)arg;
    orc_channel_t c;
    j->rc = orc_channel_open(&c, "FT8", j->fs, j->iq_len, j->demod_hz, 0.90f, 0.20f);
    if (j->rc) return NULL;
    float *iq = (float *)malloc(sizeof(float) * 2 * j->n_per_slot);
    int16_t *pcm = (int16_t *)malloc(sizeof(int16_t) * c.frame_len);
    const double tones[2] = { (double)j->demod_hz + 900.0, (double)j->demod_hz + 2100.0 };
    orc_synth_noise(j->seed, 0, j->n_per_slot, iq);
    orc_synth_add_tones(j->fs, 0, j->n_per_slot, tones, 2, 2.0e4f, iq);
    orc_channel_boundary(&c, 1, pcm, NULL, NULL, NULL);      /* discard the empty first frame */
    double acc = 0.0;
    for (int s = 0; s < j->slots; ++s) {
        for (uint64_t k = 0; k + j->iq_len <= j->n_per_slot; k += j->iq_len)
            orc_channel_push(&c, iq + 2 * k);
        uint64_t t0;
        orc_channel_boundary(&c, 2 + (uint64_t)s, pcm, &t0, NULL, NULL);
        acc += pcm[1000] + pcm[c.frame_len / 2];
    }
    j->checksum = acc;
    free(iq); free(pcm);
    orc_channel_close(&c);
    return NULL;
}

/* Runs `threads` channels in parallel, `slots` slots each; returns wall seconds (<0 on error).
 * The timed region excludes input synthesis?  No: synthesis is done once per thread before the
 * slots loop but inside the wall clock; it is <2% of one slot and amortised over `slots`. */
double orc_bench_cpu(int threads, int slots, uint64_t fs, uint32_t iq_len, uint64_t n_per_slot)
{
    if (threads < 1 || threads > 1024) return -1.0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    bench_job_t *jobs = (bench_job_t *)calloc((size_t)threads, sizeof(bench_job_t));
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int t = 0; t < threads; ++t) {
        jobs[t].fs = fs; jobs[t].iq_len = iq_len; jobs[t].n_per_slot = n_per_slot; jobs[t].slots = slots;
        jobs[t].demod_hz = -26000 + 137 * t; jobs[t].seed = 0xC0FFEEull ^ (uint64_t)t;
        pthread_create(&th[t], NULL, bench_worker, &jobs[t]);
    }
    int bad = 0;
    for (int t = 0; t < threads; ++t) { pthread_join(th[t], NULL); bad |= jobs[t].rc; }
    clock_gettime(CLOCK_MONOTONIC, &b);
    free(th); free(jobs);
    if (bad) return -1.0;
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

/* The rest of the path for bench.py's whole-path CPU figure: prepareAudio + int16 (Instance.cpp:294-338, 238-241) plus the frame memset of
 * the slot swap (:213) on `threads` threads, `reps` FT8 frames each; returns wall seconds.  (These lines sit inside Instance.cpp, which needs
 * <windows.h>: they cannot be timed as the reference's own object code, so this is the restatement above.) */
typedef struct { int reps; double checksum; } fin_job_t;
static void *fin_worker(void *arg)
{
    fin_job_t *j = (fin_job_t *)arg;
    const size_t n = orc_frame_len("FT8");
    float *src = (float *)malloc(sizeof(float) * n), *buf = (float *)malloc(sizeof(float) * n);
    int16_t *pcm = (int16_t *)malloc(sizeof(int16_t) * n);
    for (size_t k = 0; k < n; ++k) src[k] = (k < 180000) ? (float)((int)((k * 2654435761u) >> 20 & 0xFFF) - 2048) * 7.5f : 0.0f;
    double acc = 0.0;
    for (int r = 0; r < j->reps; ++r) {
        memcpy(buf, src, sizeof(float) * n);                    /* stands in for the frame the demodulator has just filled */
        float f = orc_prepare_audio(buf, n, "FT8", 0.90f, 0.20f, NULL);
        orc_to_int16(buf, n, pcm);
        memset(buf, 0, sizeof(float) * n);                      /* :213 */
        acc += f + pcm[1000 + (r & 255)];
    }
    j->checksum = acc;
    free(src); free(buf); free(pcm);
    return NULL;
}
double orc_bench_finalize(int threads, int reps)
{
    if (threads < 1 || threads > 1024 || reps < 1) return -1.0;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    fin_job_t *jobs = (fin_job_t *)calloc((size_t)threads, sizeof(fin_job_t));
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    for (int t = 0; t < threads; ++t) { jobs[t].reps = reps; pthread_create(&th[t], NULL, fin_worker, &jobs[t]); }
    for (int t = 0; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &b);
    free(th); free(jobs);
    return (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
}

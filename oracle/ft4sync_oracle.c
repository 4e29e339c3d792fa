/*
 * ft4sync_oracle.c -- TEST INFRASTRUCTURE ONLY; *** PARITY UNPINNED *** (same status as sync_oracle.h).
 *
 * CPU restatement of the coherent FT4 sync stage that upstream WSJT-X runs on every getcandidates4 candidate
 * (lib/ft4_decode.f90: the iseg/isync search; lib/ft4/ft4_downsample.f90; lib/ft4/sync4d.f90 -- recalled, not
 * vendored, not version-pinned; CWSL_DIGI only spawns jt9, source/DecoderPool.hpp:634-676):
 *   1. spectrum of the first NMAX = 72576 frame samples (one real transform per frame);
 *   2. per candidate f0: the 630 bins [-126, +503] around i0 = nint(f0/df) times a raised-cosine window, /4032,
 *      inverse 4032-point transform -> complex baseband at 12000/18 = 666.7 Hz, normalised to unit mean power;
 *   3. sync4d: correlation with the four Costas blocks 0132 1023 2310 3201 (every other sample, 64 terms each) at
 *      start samples i0, +33, +66, +99 symbols, with a frequency tweak of idf Hz; sync = sum of the four magnitudes/64;
 *   4. the search of ft4_decode: three start-sample segments, each a coarse grid (idf -12..12 step 3, start step 4)
 *      followed by a fine grid (+-4 Hz step 1, +-5 samples step 1); a segment is kept if sync >= 1.2, it is not
 *      weaker than segment 1, and 10 < f0 + idf < 4990.
 * As in sync_oracle.c the ARITHMETIC is fixed here so that the GPU kernels can reproduce it bit for bit:
 *   real transform 72576 -> pack 36288 complex = 567 x 64: 567-point DFTs as fmaf chains over a (ascending), twiddle,
 *   64-point radix-2 DIT, real unpack; inverse 4032 = 63 x 64: 63-point DFTs over the <= 10 live rows, twiddle,
 *   64-point radix-2 DIT; CMUL = (fmaf(vr,wr,-(vi*wi)), fmaf(vr,wi,vi*wr)); tables float(cos), float(sin) of double
 *   angles with exact cardinal points; reductions in the stated order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define F4_NMAX   72576
#define F4_N2     36288            /* packed complex length = 567 x 64 */
#define F4_NA     567
#define F4_NB     64
#define F4_NP     4032             /* NMAX / NDOWN */
#define F4_NSS    32               /* NSPS / NDOWN */
#define F4_KLO    (-126)           /* window support, bins relative to i0 */
#define F4_KHI    503

typedef struct { float r, i; } cf_t;

typedef struct {
    float f0_hz, f1_hz, dt_s, sync;
    int32_t ibest, idf, seg, cand;
} orc_ft4_sync_t;

static cf_t w567[F4_NA], wn2[F4_N2], w2n[F4_N2 + 1], w64[32], w63[63], w4032[F4_NP];
static float win[F4_KHI - F4_KLO + 1];
static cf_t csync[4][64], ctwk[33][64];
static unsigned char rev6[64];
static int ready = 0;

static cf_t mk(double c, double s) { cf_t v; v.r = (float)c; v.i = (float)s; return v; }

static void tables(void)
{
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < F4_NA; ++k) w567[k] = mk(cos(2.0 * pi * k / 567.0), -sin(2.0 * pi * k / 567.0));
    for (int k = 0; k < F4_N2; ++k) wn2[k] = mk(cos(2.0 * pi * k / 36288.0), -sin(2.0 * pi * k / 36288.0));
    for (int k = 0; k <= F4_N2; ++k) w2n[k] = mk(cos(2.0 * pi * k / 72576.0), -sin(2.0 * pi * k / 72576.0));
    for (int k = 0; k < 32; ++k) w64[k] = mk(cos(2.0 * pi * k / 64.0), -sin(2.0 * pi * k / 64.0));
    for (int k = 0; k < 63; ++k) w63[k] = mk(cos(2.0 * pi * k / 63.0), -sin(2.0 * pi * k / 63.0));
    for (int k = 0; k < F4_NP; ++k) w4032[k] = mk(cos(2.0 * pi * k / 4032.0), -sin(2.0 * pi * k / 4032.0));
    w567[0] = wn2[0] = w2n[0] = w64[0] = w63[0] = w4032[0] = mk(1.0, 0.0);
    w64[16] = mk(0.0, -1.0);
    /* ft4_downsample's window: 63-bin raised-cosine rise, 504 bins flat, 63-bin fall, shifted left by 126 bins */
    for (int k = F4_KLO; k <= F4_KHI; ++k) {
        const int i = k + 126;
        double w = 1.0;
        if (i < 63) w = 0.5 * (1.0 + cos(pi * (double)(62 - i) / 63.0));
        else if (i >= 567) w = 0.5 * (1.0 + cos(pi * (double)(i - 567) / 63.0));
        win[k - F4_KLO] = (float)w;
    }
    /* sync4d's reference blocks: symbol tone t at every other baseband sample: phase 2*pi*(2 t j)/32, j = 0..15 */
    static const int icos4[4][4] = {{0, 1, 3, 2}, {1, 0, 2, 3}, {2, 3, 1, 0}, {3, 2, 0, 1}};
    for (int b = 0; b < 4; ++b)
        for (int s = 0; s < 4; ++s)
            for (int j = 0; j < 16; ++j) {
                const int p = (2 * icos4[b][s] * j) % 32;
                cf_t v = mk(cos(2.0 * pi * p / 32.0), sin(2.0 * pi * p / 32.0));
                if (p == 0) v = mk(1.0, 0.0);
                if (p == 8) v = mk(0.0, 1.0);
                if (p == 16) v = mk(-1.0, 0.0);
                if (p == 24) v = mk(0.0, -1.0);
                csync[b][16 * s + j] = v;
            }
    /* frequency tweak of idf Hz at the half rate 333.33 Hz: exp(+2 pi i idf (k+1) 36/12000), k = 0..63 */
    for (int d = -16; d <= 16; ++d)
        for (int k = 0; k < 64; ++k) {
            const double a = 2.0 * pi * (double)d * (double)(k + 1) * 36.0 / 12000.0;
            ctwk[d + 16][k] = (d == 0) ? mk(1.0, 0.0) : mk(cos(a), sin(a));
        }
    for (int b = 0; b < 64; ++b) {
        int r = 0;
        for (int t = 0; t < 6; ++t) if (b & (1 << t)) r |= 1 << (5 - t);
        rev6[b] = (unsigned char)r;
    }
    ready = 1;
}

#define CMUL(v, w, t) do { (t).r = fmaf((v).r, (w).r, -((v).i * (w).i)); (t).i = fmaf((v).r, (w).i, (v).i * (w).r); } while (0)
#define CMULC(v, w, t) do { (t).r = fmaf((v).r, (w).r, (v).i * (w).i); (t).i = fmaf((v).i, (w).r, -((v).r * (w).i)); } while (0)   /* v * conj(w) */

/* 64-point radix-2 DIT in place on natural-order output (input already bit-reversed); inverse: conjugate twiddles */
static void fft64(cf_t *y, int inverse)
{
    for (int len = 2; len <= 64; len <<= 1) {
        const int half = len >> 1, step = 64 / len;
        for (int base = 0; base < 64; base += len)
            for (int k = 0; k < half; ++k) {
                cf_t t; const cf_t u = y[base + k];
                if (inverse) CMULC(y[base + k + half], w64[k * step], t); else CMUL(y[base + k + half], w64[k * step], t);
                y[base + k].r = u.r + t.r; y[base + k].i = u.i + t.i;
                y[base + k + half].r = u.r - t.r; y[base + k + half].i = u.i - t.i;
            }
    }
}

/* cx[0 .. 36288]: spectrum of the first 72576 samples (unscaled, as ft4_downsample's x = dd) */
int orc_ft4_bigspec(const int16_t *frame, float *cx_ri)
{
    if (!ready) tables();
    cf_t *cx = (cf_t *)cx_ri;
    cf_t *y = (cf_t *)malloc(sizeof(cf_t) * F4_N2);      /* y[c][b], b already bit-reversed */
    if (!y) return -1;
    for (int c = 0; c < F4_NA; ++c)
        for (int b = 0; b < F4_NB; ++b) {
            float yr = 0.0f, yi = 0.0f;
            int idx = 0;                                  /* (a * c) mod 567 */
            for (int a = 0; a < F4_NA; ++a) {
                const int m = F4_NB * a + b;
                const float zr = (float)frame[2 * m], zi = (float)frame[2 * m + 1];
                const cf_t w = w567[idx];
                yr = fmaf(zr, w.r, yr); yr = fmaf(-zi, w.i, yr);
                yi = fmaf(zr, w.i, yi); yi = fmaf(zi, w.r, yi);
                idx += c; if (idx >= F4_NA) idx -= F4_NA;
            }
            cf_t v, t; v.r = yr; v.i = yi;
            CMUL(v, wn2[b * c], t);
            y[c * F4_NB + rev6[b]] = t;
        }
    for (int c = 0; c < F4_NA; ++c) fft64(y + c * F4_NB, 0);       /* Z[c + 567 d] = y[c][d] */
    for (int k = 0; k <= F4_N2; ++k) {
        const int kk = k % F4_N2, k2 = (F4_N2 - k) % F4_N2;
        const cf_t A = y[(kk % F4_NA) * F4_NB + kk / F4_NA];
        cf_t B = y[(k2 % F4_NA) * F4_NB + k2 / F4_NA];
        B.i = -B.i;
        cf_t E, O, T;
        E.r = (A.r + B.r) * 0.5f; E.i = (A.i + B.i) * 0.5f;
        O.r = (A.r - B.r) * 0.5f; O.i = (A.i - B.i) * 0.5f;
        CMUL(O, w2n[k], T);
        cx[k].r = E.r + T.i;                              /* E + (-i) T */
        cx[k].i = E.i - T.r;
    }
    free(y);
    return 0;
}

/* cd[0 .. 4031]: normalised complex baseband around f0 (ft4_downsample + the power normalisation of ft4_decode) */
int orc_ft4_downsample(const float *cx_ri, float f0_hz, float *cd_ri)
{
    if (!ready) tables();
    const cf_t *cx = (const cf_t *)cx_ri;
    cf_t *cd = (cf_t *)cd_ri;
    const float df = 12000.0f / (float)F4_NMAX;
    const int i0 = (int)lroundf(f0_hz / df);
    static cf_t c1[F4_NP], y[63 * 64];
    for (int j = 0; j < F4_NP; ++j) { c1[j].r = 0.0f; c1[j].i = 0.0f; }
    for (int k = F4_KLO; k <= F4_KHI; ++k) {
        const int idx = i0 + k;
        if (idx < 0 || idx > F4_N2) continue;
        const float w = win[k - F4_KLO];
        cf_t v;
        v.r = (cx[idx].r * w) / 4032.0f;
        v.i = (cx[idx].i * w) / 4032.0f;
        c1[(k + F4_NP) % F4_NP] = v;
    }
    /* inverse 4032 = 63 x 64, j = 64 a + b: only rows a in {0..7, 61, 62} can be live */
    static const int live[10] = {0, 1, 2, 3, 4, 5, 6, 7, 61, 62};
    for (int c = 0; c < 63; ++c)
        for (int b = 0; b < 64; ++b) {
            float yr = 0.0f, yi = 0.0f;
            for (int q = 0; q < 10; ++q) {
                const int a = live[q];
                const cf_t z = c1[64 * a + b], w = w63[(a * c) % 63];       /* times conj(w): inverse transform */
                yr = fmaf(z.r, w.r, yr); yr = fmaf(z.i, w.i, yr);
                yi = fmaf(z.i, w.r, yi); yi = fmaf(-z.r, w.i, yi);
            }
            cf_t v, t; v.r = yr; v.i = yi;
            CMULC(v, w4032[b * c], t);
            y[c * 64 + rev6[b]] = t;
        }
    for (int c = 0; c < 63; ++c) fft64(y + c * 64, 1);
    for (int m = 0; m < F4_NP; ++m) cd[m] = y[(m % 63) * 64 + m / 63];      /* c[c + 63 d] = y[c][d] */
    /* sum2 = sum |cd|^2 / 4032: 256 strided partial sums (q ascending), then a halving tree (t, t + h) */
    float part[256];
    for (int t = 0; t < 256; ++t) {
        float s = 0.0f;
        for (int m = t; m < F4_NP; m += 256) s = fmaf(cd[m].r, cd[m].r, fmaf(cd[m].i, cd[m].i, s));
        part[t] = s;
    }
    for (int h = 128; h >= 1; h >>= 1)
        for (int t = 0; t < h; ++t) part[t] = part[t] + part[t + h];
    const float sum2 = part[0] / 4032.0f;
    if (sum2 > 0.0f) {
        const float s = sqrtf(sum2);
        for (int m = 0; m < F4_NP; ++m) { cd[m].r = cd[m].r / s; cd[m].i = cd[m].i / s; }
    }
    return i0;
}

/* z = sum_{k in [k0, k0+n)} cd[i + 2 (k - k0 ...)] ... : one Costas block, terms k = ka .. kb-1 of csync2, data start at `start` */
static cf_t corr(const cf_t *cd, int start, const cf_t *cs, int ka, int kb)
{
    cf_t z; z.r = 0.0f; z.i = 0.0f;
    for (int k = ka; k < kb; ++k) {
        const cf_t c = cd[start + 2 * (k - ka)], s = cs[k];
        z.r = fmaf(c.r, s.r, z.r); z.r = fmaf(c.i, s.i, z.r);               /* c * conj(s) */
        z.i = fmaf(c.i, s.r, z.i); z.i = fmaf(-c.r, s.i, z.i);
    }
    return z;
}
static float pmag(cf_t z)
{
    const float fac = 1.0f / 64.0f;
    const float a = z.r * fac, b = z.i * fac;
    return sqrtf(fmaf(a, a, b * b));
}

/* sync4d(cd, i0, ctwk(:, idf), 1, sync) */
float orc_ft4_sync4d(const float *cd_ri, int i0, int idf)
{
    if (!ready) tables();
    const cf_t *cd = (const cf_t *)cd_ri;
    cf_t cs[4][64];
    for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 64; ++k) CMUL(ctwk[idf + 16][k], csync[b][k], cs[b][k]);
    const int i1 = i0, i2 = i0 + 33 * F4_NSS, i3 = i0 + 66 * F4_NSS, i4 = i0 + 99 * F4_NSS;
    cf_t z1 = {0, 0}, z2 = {0, 0}, z3 = {0, 0}, z4 = {0, 0};
    if (i1 >= 0 && i1 + 4 * F4_NSS - 1 <= F4_NP - 1) z1 = corr(cd, i1, cs[0], 0, 64);
    if (i1 < 0) {
        const int npts = (i1 + 4 * F4_NSS - 1) / 2;
        if (npts > 16) z1 = corr(cd, 0, cs[0], 63 - npts, 64);            /* the last npts+1 terms on samples 0, 2, 4, ... */
    }
    if (i2 >= 0 && i2 + 4 * F4_NSS - 1 <= F4_NP - 1) z2 = corr(cd, i2, cs[1], 0, 64);
    if (i3 >= 0 && i3 + 4 * F4_NSS - 1 <= F4_NP - 1) z3 = corr(cd, i3, cs[2], 0, 64);
    if (i4 >= 0 && i4 + 4 * F4_NSS - 1 <= F4_NP - 1) z4 = corr(cd, i4, cs[3], 0, 64);
    if (i4 + 4 * F4_NSS - 1 > F4_NP - 1) {
        const int npts = (F4_NP - 1 - i4 + 1) / 2;
        if (npts > 16) z4 = corr(cd, i4, cs[3], 0, npts); else { z4.r = 0; z4.i = 0; }
    }
    return ((pmag(z1) + pmag(z2)) + pmag(z3)) + pmag(z4);
}

/* the iseg / isync search of ft4_decode for one candidate; appends up to 3 records; returns the count appended */
int orc_ft4_search(const float *cd_ri, float f0_hz, int cand, orc_ft4_sync_t *out, int max_out)
{
    int n = 0;
    float smax1 = 0.0f, smax = -99.0f;
    for (int iseg = 1; iseg <= 3; ++iseg) {
        int ibest = -1, idfbest = 0;
        for (int isync = 1; isync <= 2; ++isync) {
            int idfmin, idfmax, idfstp, ibmin, ibmax, ibstp;
            if (isync == 1) {
                idfmin = -12; idfmax = 12; idfstp = 3;
                if (iseg == 1) { ibmin = 108; ibmax = 560; }
                else if (iseg == 2) { smax1 = smax; ibmin = 560; ibmax = 1012; }
                else { ibmin = -344; ibmax = 108; }
                ibstp = 4;
            } else {
                idfmin = idfbest - 4; idfmax = idfbest + 4; idfstp = 1;
                ibmin = ibest - 5; ibmax = ibest + 5; ibstp = 1;
            }
            ibest = -1; idfbest = 0; smax = -99.0f;
            for (int idf = idfmin; idf <= idfmax; idf += idfstp)
                for (int istart = ibmin; istart <= ibmax; istart += ibstp) {
                    const float sync = orc_ft4_sync4d(cd_ri, istart, idf);
                    if (sync > smax) { smax = sync; ibest = istart; idfbest = idf; }
                }
        }
        if (iseg == 1) smax1 = smax;
        if (smax < 1.2f) continue;
        if (iseg > 1 && smax < smax1) continue;
        const float f1 = f0_hz + (float)idfbest;
        if (f1 <= 10.0f || f1 >= 4990.0f) continue;
        if (n < max_out) {
            out[n].f0_hz = f0_hz; out[n].f1_hz = f1; out[n].dt_s = (float)ibest / 666.67f - 0.5f; out[n].sync = smax;
            out[n].ibest = ibest; out[n].idf = idfbest; out[n].seg = iseg; out[n].cand = cand;
        }
        ++n;
    }
    return n < max_out ? n : max_out;
}

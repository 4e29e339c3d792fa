/* longsync_oracle.c -- TEST INFRASTRUCTURE ONLY; *** PARITY UNPINNED *** (see longsync_oracle.h).
 *
 * "spec B": the complex DFT of N = NA x NB points used by both 120 s searches (WSPR: 46080 = 45 x 1024, FST4W: 32000 = 125 x 256)
 *   input   z[n], n = NB a + b   (a < NA, b < NB)
 *   stage 1 for every column b and output c < NA, four fmaf chains over a = 0 .. NA-1 from +0:
 *             P = fmaf(zr_a, wr, P)  Q = fmaf(zi_a, wi, Q)  R = fmaf(zr_a, wi, R)  S = fmaf(zi_a, wr, S),  w = WA[(a c) mod NA]
 *             y[c][b] = cmul( (P - Q, R + S), WN[b c] )
 *   stage 2 for every row c an NB-point radix-2 DIT FFT over b (bit-reversed input): t = cmul(v, WB[k step]); (u, v) <- (u + t, u - t)
 *   output  Z[c + NA d] = y[c][d]
 *   cmul    (vr + i vi)(wr + i wi) = ( fmaf(vr, wr, -(vi*wi)),  fmaf(vr, wi, vi*wr) )
 *   tables  float(cos), float(-sin) of the double angle 2 pi k / n; W^0 = (1, 0) and W_NB^(NB/4) = (0, -1) exactly
 *   inverse conj -> forward -> conj, not normalised
 * fmaf is the correctly-rounded fused multiply-add; everything else is plain float + - * / sqrt.
 */
#include "longsync_oracle.h"
#include "sync_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CMUL(vr, vi, wr, wi, tr, ti) do { (tr) = fmaf((vr), (wr), -((vi) * (wi))); (ti) = fmaf((vr), (wi), (vi) * (wr)); } while (0)

typedef struct {
    int na, nb, n, logb;
    float *war, *wai;          /* NA  */
    float *wnr, *wni;          /* N   */
    float *wbr, *wbi;          /* NB/2 */
    int *rev;                  /* NB  */
} planb_t;

static planb_t plans[4];
static int n_plans = 0;

static const planb_t *get_plan(int na, int nb)
{
    for (int i = 0; i < n_plans; ++i) if (plans[i].na == na && plans[i].nb == nb) return &plans[i];
    if (n_plans == 4) return NULL;
    const double pi = 3.14159265358979323846;
    planb_t *P = &plans[n_plans];
    P->na = na; P->nb = nb; P->n = na * nb;
    P->logb = 0; while ((1 << P->logb) < nb) ++P->logb;
    if ((1 << P->logb) != nb) return NULL;
    P->war = (float *)malloc(sizeof(float) * na); P->wai = (float *)malloc(sizeof(float) * na);
    P->wnr = (float *)malloc(sizeof(float) * P->n); P->wni = (float *)malloc(sizeof(float) * P->n);
    P->wbr = (float *)malloc(sizeof(float) * (nb / 2)); P->wbi = (float *)malloc(sizeof(float) * (nb / 2));
    P->rev = (int *)malloc(sizeof(int) * nb);
    for (int k = 0; k < na; ++k) { P->war[k] = (float)cos(2.0 * pi * k / na); P->wai[k] = (float)(-sin(2.0 * pi * k / na)); }
    P->war[0] = 1.0f; P->wai[0] = 0.0f;
    for (int k = 0; k < P->n; ++k) { P->wnr[k] = (float)cos(2.0 * pi * k / P->n); P->wni[k] = (float)(-sin(2.0 * pi * k / P->n)); }
    P->wnr[0] = 1.0f; P->wni[0] = 0.0f;
    for (int k = 0; k < nb / 2; ++k) { P->wbr[k] = (float)cos(2.0 * pi * k / nb); P->wbi[k] = (float)(-sin(2.0 * pi * k / nb)); }
    P->wbr[0] = 1.0f; P->wbi[0] = 0.0f; P->wbr[nb / 4] = 0.0f; P->wbi[nb / 4] = -1.0f;
    for (int b = 0; b < nb; ++b) {
        int r = 0;
        for (int t = 0; t < P->logb; ++t) if (b & (1 << t)) r |= 1 << (P->logb - 1 - t);
        P->rev[b] = r;
    }
    ++n_plans;
    return P;
}

/* forward transform, out of place: (zr, zi)[n] -> (Zr, Zi)[n]; y is scratch of 2 n floats */
static void fftb_forward(const planb_t *P, const float *zr, const float *zi, float *Zr, float *Zi, float *y)
{
    const int na = P->na, nb = P->nb;
    float *yr = y, *yi = y + P->n;                       /* y[c][d] at c*nb + d */
    for (int b = 0; b < nb; ++b) {
        for (int c = 0; c < na; ++c) {
            float Ps = 0.0f, Qs = 0.0f, Rs = 0.0f, Ss = 0.0f;
            int idx = 0;
            for (int a = 0; a < na; ++a) {
                const float wr = P->war[idx], wi = P->wai[idx];
                const float xr = zr[nb * a + b], xi = zi[nb * a + b];
                Ps = fmaf(xr, wr, Ps); Qs = fmaf(xi, wi, Qs); Rs = fmaf(xr, wi, Rs); Ss = fmaf(xi, wr, Ss);
                idx += c; if (idx >= na) idx -= na;
            }
            float qr, qi;
            CMUL(Ps - Qs, Rs + Ss, P->wnr[b * c], P->wni[b * c], qr, qi);
            yr[c * nb + P->rev[b]] = qr; yi[c * nb + P->rev[b]] = qi;
        }
    }
    for (int c = 0; c < na; ++c) {
        float *rr = yr + c * nb, *ri = yi + c * nb;
        for (int len = 2; len <= nb; len <<= 1) {
            const int half = len >> 1, step = nb / len;
            for (int base = 0; base < nb; base += len) {
                for (int k = 0; k < half; ++k) {
                    const float ur = rr[base + k], ui = ri[base + k];
                    float tr, ti;
                    CMUL(rr[base + k + half], ri[base + k + half], P->wbr[k * step], P->wbi[k * step], tr, ti);
                    rr[base + k] = ur + tr;        ri[base + k] = ui + ti;
                    rr[base + k + half] = ur - tr; ri[base + k + half] = ui - ti;
                }
            }
        }
    }
    for (int c = 0; c < na; ++c)
        for (int d = 0; d < nb; ++d) { Zr[c + na * d] = yr[c * nb + d]; Zi[c + na * d] = yi[c * nb + d]; }
}

int orc_fftb(int na, int nb, float *re, float *im, int inverse)
{
    const planb_t *P = get_plan(na, nb);
    if (!P) return -1;
    const int n = P->n;
    float *buf = (float *)malloc(sizeof(float) * (size_t)n * 4);
    if (!buf) return -1;
    float *zi = buf, *y = buf + n;                       /* y: 2n, then Zi reuse */
    float *Zr = (float *)malloc(sizeof(float) * (size_t)n * 2), *Zi = Zr + n;
    for (int k = 0; k < n; ++k) zi[k] = inverse ? -im[k] : im[k];
    fftb_forward(P, re, zi, Zr, Zi, y);
    for (int k = 0; k < n; ++k) { re[k] = Zr[k]; im[k] = inverse ? -Zi[k] : Zi[k]; }
    free(buf); free(Zr);
    return 0;
}

/* ==================================================================================================== WSPR */
static const unsigned char pr3[162] = {
    1,1,0,0,0,0,0,0,1,0,0,0,1,1,1,0,0,0,1,0, 0,1,0,1,1,1,1,0,0,0,0,0,0,0,1,0,0,1,0,1,
    0,0,0,0,0,0,1,0,1,1,0,0,1,1,0,1,0,0,0,1, 1,0,1,0,0,0,0,1,1,0,1,0,1,0,1,0,1,0,0,1,
    0,0,1,0,1,1,0,0,0,1,1,0,1,0,1,0,0,0,1,0, 0,0,0,0,1,0,0,1,0,0,1,1,1,0,1,1,0,0,1,1,
    0,1,0,0,0,1,1,1,0,0,0,0,0,1,0,1,0,0,1,1, 0,0,0,0,0,0,0,1,1,0,1,0,1,1,0,0,0,1,1,0,
    0,0};

static float *wspr_tr = NULL, *wspr_ti = NULL;          /* T_a[i] = exp(-2 pi i a j_i / nfft1), [32][46080] */

static void wspr_tables(void)
{
    if (wspr_tr) return;
    const double pi = 3.14159265358979323846;
    wspr_tr = (float *)malloc(sizeof(float) * WSPR_NDEC * WSPR_NFFT2);
    wspr_ti = (float *)malloc(sizeof(float) * WSPR_NDEC * WSPR_NFFT2);
    const double df = 12000.0 / WSPR_NFFT1;
    const long i0 = (long)(1500.0 / df + 0.5);
    for (int a = 0; a < WSPR_NDEC; ++a)
        for (int i = 0; i < WSPR_NFFT2; ++i) {
            long j = i0 + i;
            if (i > WSPR_NFFT2 / 2) j -= WSPR_NFFT2;
            const long ph = ((long)a * j) % WSPR_NFFT1;
            const double ang = 2.0 * pi * (double)ph / (double)WSPR_NFFT1;
            wspr_tr[a * WSPR_NFFT2 + i] = (ph == 0) ? 1.0f : (float)cos(ang);
            wspr_ti[a * WSPR_NFFT2 + i] = (ph == 0) ? 0.0f : (float)(-sin(ang));
        }
}

/* what wsprd reads as sample n of the file the reference writes (46-byte header, 44 skipped) */
static inline float wspr_sample(const int16_t *frame, int frame_len, int n)
{
    if (n >= WSPR_NPTS) return 0.0f;
    int v;
    if (n == 0) v = (int)(int16_t)(((unsigned)frame_len * 2u) >> 16);     /* upper half of the data-length field */
    else v = (n - 1 < frame_len) ? frame[n - 1] : 0;
    return (float)v * (1.0f / 32768.0f);
}

int orc_wspr_downsample(const int16_t *frame, int frame_len, float *idat, float *qdat)
{
    const planb_t *P = get_plan(45, 1024);
    if (!P) return -1;
    wspr_tables();
    const int M = WSPR_NFFT2;
    float *Yr = (float *)malloc(sizeof(float) * (size_t)M * WSPR_NDEC * 2), *Yi = Yr + (size_t)M * WSPR_NDEC;
    float *w = (float *)malloc(sizeof(float) * (size_t)M * 6);
    if (!Yr || !w) { free(Yr); free(w); return -1; }
    float *zr = w, *zi = w + M, *Zr = w + 2 * M, *Zi = w + 3 * M, *y = w + 4 * M;
    for (int p = 0; p < WSPR_NDEC / 2; ++p) {
        for (int b = 0; b < M; ++b) {
            zr[b] = wspr_sample(frame, frame_len, WSPR_NDEC * b + p);
            zi[b] = wspr_sample(frame, frame_len, WSPR_NDEC * b + p + WSPR_NDEC / 2);
        }
        fftb_forward(P, zr, zi, Zr, Zi, y);
        float *ar = Yr + (size_t)p * M, *ai = Yi + (size_t)p * M;
        float *br = Yr + (size_t)(p + WSPR_NDEC / 2) * M, *bi = Yi + (size_t)(p + WSPR_NDEC / 2) * M;
        for (int i = 0; i < M; ++i) {
            const int m = (M - i) % M;
            ar[i] = (Zr[i] + Zr[m]) * 0.5f; ai[i] = (Zi[i] - Zi[m]) * 0.5f;
            br[i] = (Zi[i] + Zi[m]) * 0.5f; bi[i] = (Zr[m] - Zr[i]) * 0.5f;
        }
    }
    /* fftin[i] = X[j_i] = sum_a Y_a[i] T_a[i]; the inverse transform by conjugation */
    for (int i = 0; i < M; ++i) {
        float fr = 0.0f, fi = 0.0f;
        for (int a = 0; a < WSPR_NDEC; ++a) {
            float tr, ti;
            CMUL(Yr[(size_t)a * M + i], Yi[(size_t)a * M + i], wspr_tr[a * M + i], wspr_ti[a * M + i], tr, ti);
            fr = fr + tr; fi = fi + ti;
        }
        zr[i] = fr; zi[i] = -fi;
    }
    fftb_forward(P, zr, zi, Zr, Zi, y);
    for (int i = 0; i < M; ++i) {
        idat[i] = (float)((double)Zr[i] / 1000.0);
        qdat[i] = (float)((double)(-Zi[i]) / 1000.0);
    }
    free(Yr); free(w);
    return 0;
}

/* 512-point radix-2 DIT, same butterfly as spec B's stage 2 */
static float w512r[256], w512i[256];
static int rev9[512];
static float wwin[512];
static int t512 = 0;
static void tables512(void)
{
    if (t512) return;
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < 256; ++k) { w512r[k] = (float)cos(2.0 * pi * k / 512.0); w512i[k] = (float)(-sin(2.0 * pi * k / 512.0)); }
    w512r[0] = 1.0f; w512i[0] = 0.0f; w512r[128] = 0.0f; w512i[128] = -1.0f;
    for (int b = 0; b < 512; ++b) { int r = 0; for (int t = 0; t < 9; ++t) if (b & (1 << t)) r |= 1 << (8 - t); rev9[b] = r; }
    for (int j = 0; j < 512; ++j) wwin[j] = (float)sin(0.006147931 * (double)j);
    t512 = 1;
}
static void fft512(float *rr, float *ri)        /* in: bit-reversed order already applied by the caller */
{
    for (int len = 2; len <= 512; len <<= 1) {
        const int half = len >> 1, step = 512 / len;
        for (int base = 0; base < 512; base += len)
            for (int k = 0; k < half; ++k) {
                const float ur = rr[base + k], ui = ri[base + k];
                float tr, ti;
                CMUL(rr[base + k + half], ri[base + k + half], w512r[k * step], w512i[k * step], tr, ti);
                rr[base + k] = ur + tr;        ri[base + k] = ui + ti;
                rr[base + k + half] = ur - tr; ri[base + k + half] = ui - ti;
            }
    }
}

static int cmp_float(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return (x > y) - (x < y);
}

int orc_wspr_search(const int16_t *frame, int frame_len, orc_wspr_cand_t *out, int max_out,
                    float *idat_o, float *qdat_o, float *ps_o, float *smspec_o)
{
    const int nffts = WSPR_NFFTS;
    float *idat = (float *)calloc(65536, sizeof(float)), *qdat = (float *)calloc(65536, sizeof(float));
    float *ps = (float *)malloc(sizeof(float) * 512 * nffts), *sq = (float *)malloc(sizeof(float) * 512 * nffts);
    if (!idat || !qdat || !ps || !sq || orc_wspr_downsample(frame, frame_len, idat, qdat)) { free(idat); free(qdat); free(ps); free(sq); return -1; }
    tables512();
    if (idat_o) memcpy(idat_o, idat, sizeof(float) * WSPR_NFFT2);
    if (qdat_o) memcpy(qdat_o, qdat, sizeof(float) * WSPR_NFFT2);
    const float df = (float)(375.0 / 256.0 / 2);
    for (int i = 0; i < nffts; ++i) {
        float fr[512], fi[512];
        for (int j = 0; j < 512; ++j) {
            const int k = i * 128 + j;                   /* idat/qdat are calloc'ed to 65536: zero beyond 46080 */
            fr[rev9[j]] = idat[k] * wwin[j];
            fi[rev9[j]] = qdat[k] * wwin[j];
        }
        fft512(fr, fi);
        for (int j = 0; j < 512; ++j) {
            int k = j + 256; if (k > 511) k -= 512;
            const float p = fr[k] * fr[k] + fi[k] * fi[k];
            ps[j * nffts + i] = p;
            sq[j * nffts + i] = sqrtf(p);
        }
    }
    if (ps_o) memcpy(ps_o, ps, sizeof(float) * 512 * nffts);
    float psavg[512];
    for (int j = 0; j < 512; ++j) { float s = 0.0f; for (int i = 0; i < nffts; ++i) s = s + ps[j * nffts + i]; psavg[j] = s; }
    float smspec[411], tmpsort[411];
    for (int i = 0; i < 411; ++i) {
        float s = 0.0f;
        for (int j = -3; j <= 3; ++j) s = s + psavg[256 - 205 + i + j];
        smspec[i] = s;
        tmpsort[i] = s;
    }
    qsort(tmpsort, 411, sizeof(float), cmp_float);
    const float noise_level = tmpsort[122];
    const float min_snr = (float)pow(10.0, -8.0 / 10.0);
    const float snr_scaling_factor = 26.3f;
    for (int j = 0; j < 411; ++j) {
        smspec[j] = (float)((double)(smspec[j] / noise_level) - 1.0);
        if (smspec[j] < min_snr) smspec[j] = (float)(0.1 * (double)min_snr);
    }
    if (smspec_o) memcpy(smspec_o, smspec, sizeof(smspec));
    float freq0[200], snr0[200], drift0[200], sync0[200];
    int shift0[200];
    int npk = 0;
    for (int j = 1; j < 410; ++j) {
        if (smspec[j] > smspec[j - 1] && smspec[j] > smspec[j + 1] && npk < 200) {
            freq0[npk] = (float)(j - 205) * df;
            snr0[npk] = (float)(10.0 * orc_log10_fixed((double)smspec[j]) - (double)snr_scaling_factor);
            ++npk;
        }
    }
    const float fmin = -110.0f, fmax = 110.0f;
    int n2 = 0;
    for (int j = 0; j < npk; ++j)
        if (freq0[j] >= fmin && freq0[j] <= fmax) { freq0[n2] = freq0[j]; snr0[n2] = snr0[j]; ++n2; }
    npk = n2;
    for (int pass = 1; pass <= npk - 1; ++pass)
        for (int k = 0; k < npk - pass; ++k)
            if (snr0[k] < snr0[k + 1]) {
                float t = snr0[k]; snr0[k] = snr0[k + 1]; snr0[k + 1] = t;
                t = freq0[k]; freq0[k] = freq0[k + 1]; freq0[k + 1] = t;
            }
    const int maxdrift = 4;
    for (int j = 0; j < npk; ++j) {
        float smax = -1e30f;
        drift0[j] = 0.0f; shift0[j] = 0; sync0[j] = 0.0f;
        const int if0 = (int)(freq0[j] / df + 256.0f);
        for (int ifr = if0 - 2; ifr <= if0 + 2; ++ifr)
            for (int k0 = -10; k0 < 22; ++k0)
                for (int idrift = -maxdrift; idrift <= maxdrift; ++idrift) {
                    float ss = 0.0f, pw = 0.0f;
                    for (int k = 0; k < 162; ++k) {
                        const int ifd = (int)((double)ifr + ((double)(float)k - 81.0) / 81.0 * (double)(float)idrift / (2.0 * (double)df));
                        const int kindex = k0 + 2 * k;
                        if (kindex < nffts) {
                            /* flat indexing: a negative kindex lands at the end of the previous row, as in wsprd */
                            const float p0 = sq[(ifd - 3) * nffts + kindex], p1 = sq[(ifd - 1) * nffts + kindex];
                            const float p2 = sq[(ifd + 1) * nffts + kindex], p3 = sq[(ifd + 3) * nffts + kindex];
                            ss = ss + (float)(2 * pr3[k] - 1) * ((p1 + p3) - (p0 + p2));
                            pw = pw + p0 + p1 + p2 + p3;
                        }
                    }
                    const float sync1 = ss / pw;
                    if (sync1 > smax) {
                        smax = sync1;
                        shift0[j] = 128 * (k0 + 1);
                        drift0[j] = (float)idrift;
                        freq0[j] = (float)(ifr - 256) * df;
                        sync0[j] = sync1;
                    }
                }
    }
    int n = 0;
    for (int j = 0; j < npk && n < max_out; ++j, ++n) {
        out[n].freq_hz = freq0[j]; out[n].snr_db = snr0[j]; out[n].drift = drift0[j]; out[n].sync = sync0[j]; out[n].shift = shift0[j];
    }
    free(idat); free(qdat); free(ps); free(sq);
    return n;
}

/* ==================================================================================================== FST4W-120
 * fst4_decode.f90: r_data = iwave (unscaled), real FFT of nfft1 = 1 440 000 points -> c_bigfft; get_candidates_fst4:
 *   df1 = fs/nfft1, baud = fs/nsps, df2 = baud/2, nd = int(df2/df1), ndh = nd/2
 *   s(i)  = sum_{j = j0-ndh}^{j0+ndh} |c_bigfft(j)|^2,  j0 = nint(i df2/df1),  i = ina .. inb  (ina/inb = nint(nfa/df2), nint(nfb/df2))
 *   s2(i) = s(i-3h) + s(i-h) + s(i+h) + s(i+3h)   (hmod h = 1),  divided by its 30th percentile over [ina+3h, inb-3h]
 *   CLEAN: repeatedly take the maximum of s2(ia:ib); stop below minsync or at 100; subtract 0.9 pval xdb(i) at iploc + 2 h i, i = -3..3
 * The search window is the -L/-H pair the reference passes (1400..1600 Hz) for both the signal and the noise window.
 * The band of c_bigfft is evaluated as the polyphase band DFT of the header with R = 45, M = 32000 = 125 x 256 (spec B).
 */
#define F4W_R 45
#define F4W_M 32000

int orc_fst4w_candidates(const int16_t *frame, int frame_len, int nfa_hz, int nfb_hz, float minsync,
                         orc_fst4w_cand_t *out, int max_out, float *s2_o, int n_s2, float *band_o)
{
    const planb_t *P = get_plan(125, 256);
    if (!P) return -1;
    const double pi = 3.14159265358979323846;
    const int hmod = 1;
    const float fs = 12000.0f;
    const int nfft1 = FST4W_NMAX, nsps = FST4W_NSPS;
    const float df1 = fs / (float)nfft1, baud = fs / (float)nsps, df2 = baud / 2.0f;
    const int nd = (int)(df2 / df1), ndh = nd / 2;
    const float fa = (float)nfa_hz, fb = (float)nfb_hz;
    int ia = (int)lroundf(fmaxf(100.0f, fa) / df2), ib = (int)lroundf(fminf(4800.0f, fb) / df2);
    int ina = (int)lroundf(fmaxf(100.0f, (float)nfa_hz) / df2), inb = (int)lroundf(fminf(4800.0f, (float)nfb_hz) / df2);
    if (ia < ina) ia = ina;
    if (ib > inb) ib = inb;
    const int nnw = (int)lroundf(48000.0f * (float)nsps * 2.0f / fs);
    if (inb < ina || inb >= nnw) return -1;
    const int jlo = (int)lroundf((float)ina * df2 / df1) - ndh, jhi = (int)lroundf((float)inb * df2 / df1) + ndh;
    const int nband = jhi - jlo + 1;
    if (jlo < 0 || jhi > nfft1 / 2 || nband > F4W_M) return -1;
    const int M = F4W_M;
    /* Y_a, a < 45: M-point DFTs of x[45 b + a]; pairs (p, p + 22) share one complex transform, a = 44 goes alone */
    float *Yr = (float *)malloc(sizeof(float) * (size_t)M * F4W_R * 2), *Yi = Yr + (size_t)M * F4W_R;
    float *w = (float *)malloc(sizeof(float) * (size_t)M * 6);
    if (!Yr || !w) { free(Yr); free(w); return -1; }
    float *zr = w, *zi = w + M, *Zr = w + 2 * M, *Zi = w + 3 * M, *y = w + 4 * M;
#define XS(n) (((n) < frame_len && (n) < nfft1) ? (float)frame[(n)] : 0.0f)
    for (int p = 0; p < 23; ++p) {
        const int a2 = (p < 22) ? p + 22 : -1;
        const int a1 = (p < 22) ? p : 44;
        for (int b = 0; b < M; ++b) {
            zr[b] = XS(F4W_R * b + a1);
            zi[b] = (a2 >= 0) ? XS(F4W_R * b + a2) : 0.0f;
        }
        fftb_forward(P, zr, zi, Zr, Zi, y);
        float *ar = Yr + (size_t)a1 * M, *ai = Yi + (size_t)a1 * M;
        for (int i = 0; i < M; ++i) {
            const int m = (M - i) % M;
            ar[i] = (Zr[i] + Zr[m]) * 0.5f; ai[i] = (Zi[i] - Zi[m]) * 0.5f;
        }
        if (a2 >= 0) {
            float *br = Yr + (size_t)a2 * M, *bi = Yi + (size_t)a2 * M;
            for (int i = 0; i < M; ++i) {
                const int m = (M - i) % M;
                br[i] = (Zi[i] + Zi[m]) * 0.5f; bi[i] = (Zr[m] - Zr[i]) * 0.5f;
            }
        }
    }
#undef XS
    /* |c_bigfft(j)|^2 for j = jlo .. jhi: X[j] = sum_a Y_a[j mod M] exp(-2 pi i a j / nfft1), a ascending */
    float *pw = (float *)malloc(sizeof(float) * (size_t)nband * 2), *pw_i = pw + nband;
    for (int q = 0; q < nband; ++q) {
        const long j = jlo + q;
        const int i = (int)(j % M);
        float fr = 0.0f, fi = 0.0f;
        for (int a = 0; a < F4W_R; ++a) {
            const long ph = ((long)a * j) % nfft1;
            const double ang = 2.0 * pi * (double)ph / (double)nfft1;
            const float tr0 = (ph == 0) ? 1.0f : (float)cos(ang), ti0 = (ph == 0) ? 0.0f : (float)(-sin(ang));
            float tr, ti;
            CMUL(Yr[(size_t)a * M + i], Yi[(size_t)a * M + i], tr0, ti0, tr, ti);
            fr = fr + tr; fi = fi + ti;
        }
        pw[q] = fr; pw_i[q] = fi;
    }
    if (band_o) for (int q = 0; q < nband; ++q) band_o[q] = pw[q] * pw[q] + pw_i[q] * pw_i[q];
    float *s = (float *)calloc((size_t)nnw + 8, sizeof(float)), *s2 = (float *)calloc((size_t)nnw + 8, sizeof(float));
    for (int i = ina; i <= inb; ++i) {
        const int j0 = (int)lroundf((float)i * df2 / df1);
        float acc = 0.0f;
        for (int j = j0 - ndh; j <= j0 + ndh; ++j) {
            const float re = pw[j - jlo], im = pw_i[j - jlo];
            acc = acc + re * re + im * im;                /* (acc + re^2) + im^2 */
        }
        s[i] = acc;
    }
    const int ina0 = ina, inb0 = inb;
    if (ina < 1 + 3 * hmod) ina = 1 + 3 * hmod;
    if (inb > nnw - 3 * hmod) inb = nnw - 3 * hmod;
    for (int i = ina; i <= inb; ++i) s2[i] = s[i - hmod * 3] + s[i - hmod] + s[i + hmod] + s[i + hmod * 3];
    (void)ina0; (void)inb0;
    /* pctile(s2(ina+3h : inb-3h), npts, 30): j = nint(npts*0.01*30), clamped to [1, npts], 1-based in the sorted copy */
    {
        const int lo = ina + hmod * 3, npts = inb - ina + 1 - hmod * 6;
        if (npts < 1) { free(Yr); free(w); free(pw); free(s); free(s2); return 0; }
        float *tmp = (float *)malloc(sizeof(float) * (size_t)npts);
        memcpy(tmp, s2 + lo, sizeof(float) * (size_t)npts);
        qsort(tmp, (size_t)npts, sizeof(float), cmp_float);
        int jp = (int)lroundf((float)npts * 0.01f * 30.0f);
        if (jp < 1) jp = 1;
        if (jp > npts) jp = npts;
        const float base = tmp[jp - 1];
        free(tmp);
        for (int i = 0; i < nnw; ++i) s2[i] = s2[i] / base;
    }
    if (s2_o) for (int i = 0; i < n_s2; ++i) s2_o[i] = (i < nnw) ? s2[i] : 0.0f;
    if (ia < 3) ia = 3;
    if (ib > nnw - 2) ib = nnw - 2;
    static const float xdb[7] = {0.25f, 0.50f, 0.75f, 1.0f, 0.75f, 0.50f, 0.25f};
    int ncand = 0;
    while (ncand < FST4W_MAXCAND && ncand < max_out) {
        int ip = ia;
        for (int i = ia + 1; i <= ib; ++i) if (s2[i] > s2[ip]) ip = i;          /* maxloc: first maximum */
        const float pval = s2[ip];
        if (pval < minsync) break;
        for (int i = -3; i <= 3; ++i) {
            const int k = ip + 2 * hmod * i;
            if (k >= ia && k <= ib) {
                const float v = s2[k] - 0.9f * pval * xdb[i + 3];
                s2[k] = (v > 0.0f) ? v : 0.0f;
            }
        }
        out[ncand].freq_hz = df2 * (float)ip;
        out[ncand].snr = pval;
        out[ncand].bin = ip;
        out[ncand].pad_ = 0;
        ++ncand;
    }
    free(Yr); free(w); free(pw); free(s); free(s2);
    return ncand;
}

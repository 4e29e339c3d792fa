/* sync_oracle.c -- TEST INFRASTRUCTURE ONLY; *** PARITY UNPINNED *** (see sync_oracle.h). */
#include "sync_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NB   128               /* radix-2 part of both transforms */

/*
 * Transform of record ("spec v3", round 4), shared by FT8 (2*NZ = 3840, NA = 15) and FT4 (2*NZ = 2304, NA = 9):
 *   input    FT8: x = the int16 samples as floats, the 1/300 of sync8 rides in stage 1's twiddle (WN' = fl(fac * WN), output 0 takes a plain
 *            multiplication by fac); FT4: x = (fac * d) * window
 *   pack     z[m] = x[2m] + i x[2m+1]  (m < NZ; FT8: zero for m >= 960), m = 128 a + b
 *   stage 1  for each column b an NA-point DFT over a, then y_c *= WN[b c] for c >= 1 (cmul below).
 *            NA = 9: evaluated in conjugate pairs (c, NA-c):
 *              P = sum_a zr_a wr, Q = sum_a zi_a wi, R = sum_a zr_a wi, S = sum_a zi_a wr   (a = 1.., fmaf chains from 0)
 *              y_c = (z0r + (P - Q), z0i + (R + S)),  y_{NA-c} = (z0r + (P + Q), z0i + (S - R)),  w = WA[(a c) mod NA]
 *              y_0 = z_0 + z_1 + ... (sequential)
 *            NA = 15 (spec v3): Good-Thomas 3 x 5 on the EIGHT live inputs a = 0..7 (a column with seven takes z_7 = 0):
 *              a = (5 n1 + 3 n2) mod 15, c = (10 k1 + 6 k2) mod 15, so W15^(a c) = W3^(n1 k1) W5^(n2 k2).
 *              five-point step, per n1, over the live n2 in ascending order (n1 = 0: n2 = 0,1,2 = a 0,3,6; n1 = 1: n2 = 0,4 = a 5,2;
 *              n1 = 2: n2 = 2,3,4 = a 1,4,7):  Y[n1][0] = sequential sum;  for k2 = 1, 2, over the inputs with n2 != 0, w = W5^(n2 k2):
 *                P = zr wr, Q = zi wi, R = zr wi, S = zi wr for the first of them (plain products), fmaf chains for the others;
 *                Y[n1][k2] = d + (P - Q, R + S),  Y[n1][5 - k2] = d + (P + Q, S - R),  d = the n2 = 0 input (absent for n1 = 2: no addition)
 *              three-point step, per k2:  X[k1 = 0] = (Y0 + Y1) + Y2;  t = Y1 + Y2, d = Y1 - Y2, m = fmaf(t, -0.5, Y0) per component,
 *                X[k1 = 1] = (fmaf(-s3, d.i, m.r), fmaf(s3, d.r, m.i)),  X[k1 = 2] = (fmaf(s3, d.i, m.r), fmaf(-s3, d.r, m.i)),  s3 = float(-sin(2 pi / 3))
 *   stage 2  NA radix-2 DIT FFTs of 128 points (input bit-reversed); butterfly (u, v, w) -> (a, d), three fmaf-class operations per component:
 *              a.r = fmaf(v.r, w.r, fmaf(-v.i, w.i, u.r)),  a.i = fmaf(v.r, w.i, fmaf(v.i, w.r, u.i)),  d = fmaf(2, u, -a)
 *            (in the first three stages, len = 2, 4, 8, the butterflies with w = (1, 0) and w = (0, -1) -- exact table entries -- are plain
 *            additions: a = u + v, d = u - v; resp. a = u + (v.i, -v.r), d = u - (v.i, -v.r); from len = 16 on every butterfly takes the fmaf form)
 *   stage 3  real-input unpack with WH[k] = 0.5 * W2N[k] (exact scaling):  A = Z[k], B = conj Z[N - k]:
 *              s = A + B, o = A - B, t = cmul(o, WH[k]),  xr = fmaf(s.r, 0.5, t.i), xi = fmaf(s.i, 0.5, -t.r),  pw = fmaf(xr, xr, xi*xi)
 *   cmul     (vr + i vi)(wr + i wi) = ( fmaf(vr, wr, -(vi*wi)),  fmaf(vr, wi, vi*wr) )
 * Twiddle tables: float(cos), float(-sin) of the double angle, EXCEPT that the cardinal points are exact
 * (W^0 = (1, 0), W128^32 = (0, -1)) and WA / W5 are conjugate-symmetric bit for bit (WA[NA-k] = conj(WA[k])).
 * fmaf is the correctly-rounded fused multiply-add (C99); everything else is plain float + - *.
 * (Spec v2, rounds 2-3: the NA = 9 form of stage 1 for both modes, t = cmul(v, w) and (u + t, u - t) butterflies, halving before the unpack
 * product.  v3 is the same transform with about 17 % fewer operations; both agree with a double-precision FFT to the same 2e-6.)
 */
typedef struct {
    int na, nz, npack;
    float war[16], wai[16];
    float *wnr, *wni;          /* exp(-2 pi i k/NZ),  k < NZ   */
    float *w2r, *w2i;          /* exp(-2 pi i k/2NZ), k <= NZ  */
} fft_plan;

static float w128r[NB / 2], w128i[NB / 2]; /* exp(-2 pi i k/128)  */
static float w5r[5], w5i[5], w3s;          /* exp(-2 pi i k/5) (conjugate-symmetric), -sin(2 pi/3) */
static unsigned char rev7[NB];
static fft_plan plan8, plan4;
static int tables_ready = 0;

static void make_plan(fft_plan *P, int na, int npack)
{
    const double pi = 3.14159265358979323846;
    P->na = na; P->nz = na * NB; P->npack = npack;
    for (int k = 0; k <= na / 2; ++k) { P->war[k] = (float)cos(2.0 * pi * k / na); P->wai[k] = (float)(-sin(2.0 * pi * k / na)); }
    P->war[0] = 1.0f; P->wai[0] = 0.0f;
    for (int k = na / 2 + 1; k < na; ++k) { P->war[k] = P->war[na - k]; P->wai[k] = -P->wai[na - k]; }
    P->wnr = (float *)malloc(sizeof(float) * P->nz); P->wni = (float *)malloc(sizeof(float) * P->nz);
    P->w2r = (float *)malloc(sizeof(float) * (P->nz + 1)); P->w2i = (float *)malloc(sizeof(float) * (P->nz + 1));
    for (int k = 0; k < P->nz; ++k) { P->wnr[k] = (float)cos(2.0 * pi * k / P->nz); P->wni[k] = (float)(-sin(2.0 * pi * k / P->nz)); }
    for (int k = 0; k <= P->nz; ++k) { P->w2r[k] = (float)cos(pi * k / P->nz); P->w2i[k] = (float)(-sin(pi * k / P->nz)); }
    P->wnr[0] = 1.0f; P->wni[0] = 0.0f; P->w2r[0] = 1.0f; P->w2i[0] = 0.0f;
    for (int k = 0; k <= P->nz; ++k) { P->w2r[k] = 0.5f * P->w2r[k]; P->w2i[k] = 0.5f * P->w2i[k]; }      /* WH = 0.5 * W2N, exact */
    if (na == 15) {                                      /* FT8: the input scale 1/300 folded into the twiddle behind stage 1 */
        const float fac = 1.0f / 300.0f;
        for (int k = 0; k < P->nz; ++k) { P->wnr[k] = fac * P->wnr[k]; P->wni[k] = fac * P->wni[k]; }
    }
}

static void make_tables(void)
{
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < NB / 2; ++k) { w128r[k] = (float)cos(2.0 * pi * k / 128.0); w128i[k] = (float)(-sin(2.0 * pi * k / 128.0)); }
    w128r[0] = 1.0f; w128i[0] = 0.0f; w128r[32] = 0.0f; w128i[32] = -1.0f;
    for (int k = 0; k <= 2; ++k) { w5r[k] = (float)cos(2.0 * pi * k / 5.0); w5i[k] = (float)(-sin(2.0 * pi * k / 5.0)); }
    w5r[0] = 1.0f; w5i[0] = 0.0f;
    for (int k = 3; k < 5; ++k) { w5r[k] = w5r[5 - k]; w5i[k] = -w5i[5 - k]; }
    w3s = (float)(-sin(2.0 * pi / 3.0));
    for (int b = 0; b < NB; ++b) {
        int r = 0;
        for (int t = 0; t < 7; ++t) if (b & (1 << t)) r |= 1 << (6 - t);
        rev7[b] = (unsigned char)r;
    }
    make_plan(&plan8, 15, 960);
    make_plan(&plan4, 9, 1152);
    tables_ready = 1;
}

#define CMUL(vr, vi, wr, wi, tr, ti) do { (tr) = fmaf((vr), (wr), -((vi) * (wi))); (ti) = fmaf((vr), (wi), (vi) * (wr)); } while (0)

/* spec v3 stage 1 for NA = 15: the 15-point DFT of z_0..z_7 (z_8.. = 0) by the prime-factor algorithm; out[c], c = 0..14 */
static void dft15_pfa8(const float *zr, const float *zi, float *outr, float *outi)
{
    static const int cnt[3] = {3, 2, 3};
    static const int n2s[3][3] = {{0, 1, 2}, {0, 4, 0}, {2, 3, 4}};           /* live n2 per n1, ascending            */
    static const int as_[3][3] = {{0, 3, 6}, {5, 2, 0}, {1, 4, 7}};           /* a = (5 n1 + 3 n2) mod 15 of each one  */
    float Yr[3][5], Yi[3][5];
    for (int n1 = 0; n1 < 3; ++n1) {
        const int m = cnt[n1], dc = (n2s[n1][0] == 0);
        float sr = zr[as_[n1][0]], si = zi[as_[n1][0]];
        for (int i = 1; i < m; ++i) { sr = sr + zr[as_[n1][i]]; si = si + zi[as_[n1][i]]; }
        Yr[n1][0] = sr; Yi[n1][0] = si;
        for (int k2 = 1; k2 <= 2; ++k2) {
            float Ps = 0.0f, Qs = 0.0f, Rs = 0.0f, Ss = 0.0f;
            for (int i = dc; i < m; ++i) {
                const float xr = zr[as_[n1][i]], xi = zi[as_[n1][i]];
                const float wr = w5r[(n2s[n1][i] * k2) % 5], wi = w5i[(n2s[n1][i] * k2) % 5];
                if (i == dc) { Ps = xr * wr; Qs = xi * wi; Rs = xr * wi; Ss = xi * wr; }
                else { Ps = fmaf(xr, wr, Ps); Qs = fmaf(xi, wi, Qs); Rs = fmaf(xr, wi, Rs); Ss = fmaf(xi, wr, Ss); }
            }
            if (dc) {
                const float dr = zr[as_[n1][0]], di = zi[as_[n1][0]];
                Yr[n1][k2] = dr + (Ps - Qs);     Yi[n1][k2] = di + (Rs + Ss);
                Yr[n1][5 - k2] = dr + (Ps + Qs); Yi[n1][5 - k2] = di + (Ss - Rs);
            } else {
                Yr[n1][k2] = Ps - Qs;     Yi[n1][k2] = Rs + Ss;
                Yr[n1][5 - k2] = Ps + Qs; Yi[n1][5 - k2] = Ss - Rs;
            }
        }
    }
    for (int k2 = 0; k2 < 5; ++k2) {
        const float y0r = Yr[0][k2], y0i = Yi[0][k2], y1r = Yr[1][k2], y1i = Yi[1][k2], y2r = Yr[2][k2], y2i = Yi[2][k2];
        outr[(6 * k2) % 15] = (y0r + y1r) + y2r;  outi[(6 * k2) % 15] = (y0i + y1i) + y2i;
        const float tr = y1r + y2r, ti = y1i + y2i, dr = y1r - y2r, di = y1i - y2i;
        const float mr = fmaf(tr, -0.5f, y0r), mi = fmaf(ti, -0.5f, y0i);
        outr[(10 + 6 * k2) % 15] = fmaf(-w3s, di, mr);  outi[(10 + 6 * k2) % 15] = fmaf(w3s, dr, mi);
        outr[(20 + 6 * k2) % 15] = fmaf(w3s, di, mr);   outi[(20 + 6 * k2) % 15] = fmaf(-w3s, dr, mi);
    }
}

/* x[0 .. 2*npack) real (implicitly zero padded to 2*NZ); pw[k] = |X[k]|^2 for k in [0, nbins), nbins <= NZ + 1 */
static void spectrum_packed(const fft_plan *P, const float *x, float *pw, int nbins)
{
    float yr[16][NB], yi[16][NB];                         /* (automatic: the restatement is called from several test threads at once) */
    const int na = P->na, nz = P->nz;
    for (int b = 0; b < NB; ++b) {
        int amax = 0;
        while (amax < na && NB * amax + b < P->npack) ++amax;              /* inputs a < amax are live */
        if (na == 15) {
            float zr[8], zi[8], cr[15], ci[15];
            for (int a = 0; a < 8; ++a) { zr[a] = (a < amax) ? x[2 * (NB * a + b)] : 0.0f; zi[a] = (a < amax) ? x[2 * (NB * a + b) + 1] : 0.0f; }
            dft15_pfa8(zr, zi, cr, ci);
            yr[0][rev7[b]] = cr[0] * P->wnr[0]; yi[0][rev7[b]] = ci[0] * P->wnr[0];     /* c = 0: WN'^0 = (fac, 0): plain products */
            for (int c = 1; c < 15; ++c) {
                float qr, qi;
                CMUL(cr[c], ci[c], P->wnr[b * c], P->wni[b * c], qr, qi);
                yr[c][rev7[b]] = qr; yi[c][rev7[b]] = qi;
            }
            continue;
        }
        const float z0r = x[2 * b], z0i = x[2 * b + 1];
        float s0r = z0r, s0i = z0i;
        for (int a = 1; a < amax; ++a) { s0r = s0r + x[2 * (NB * a + b)]; s0i = s0i + x[2 * (NB * a + b) + 1]; }
        yr[0][rev7[b]] = s0r; yi[0][rev7[b]] = s0i;                         /* c = 0: WN^0 = 1, no multiply */
        for (int c = 1; c <= na / 2; ++c) {
            float Ps = 0.0f, Qs = 0.0f, Rs = 0.0f, Ss = 0.0f;
            for (int a = 1; a < amax; ++a) {
                const float zr = x[2 * (NB * a + b)], zi = x[2 * (NB * a + b) + 1];
                const float wr = P->war[(a * c) % na], wi = P->wai[(a * c) % na];
                Ps = fmaf(zr, wr, Ps); Qs = fmaf(zi, wi, Qs); Rs = fmaf(zr, wi, Rs); Ss = fmaf(zi, wr, Ss);
            }
            const float ar = z0r + (Ps - Qs), ai = z0i + (Rs + Ss);         /* output c      */
            const float br = z0r + (Ps + Qs), bi = z0i + (Ss - Rs);         /* output na - c */
            float qr, qi;
            CMUL(ar, ai, P->wnr[b * c], P->wni[b * c], qr, qi);
            yr[c][rev7[b]] = qr; yi[c][rev7[b]] = qi;
            CMUL(br, bi, P->wnr[b * (na - c)], P->wni[b * (na - c)], qr, qi);
            yr[na - c][rev7[b]] = qr; yi[na - c][rev7[b]] = qi;
        }
    }
    for (int c = 0; c < na; ++c) {
        for (int len = 2; len <= NB; len <<= 1) {
            const int half = len >> 1, step = NB / len;
            for (int base = 0; base < NB; base += len) {
                for (int k = 0; k < half; ++k) {
                    const float ur = yr[c][base + k], ui = yi[c][base + k];
                    const float vr = yr[c][base + k + half], vi = yi[c][base + k + half];
                    const int kw = k * step;
                    if (len <= 8 && kw == 0) {                        /* w = (1, 0)  */
                        yr[c][base + k] = ur + vr;        yi[c][base + k] = ui + vi;
                        yr[c][base + k + half] = ur - vr; yi[c][base + k + half] = ui - vi;
                    } else if (len <= 8 && kw == NB / 4) {            /* w = (0, -1): t = (v.i, -v.r) */
                        yr[c][base + k] = ur + vi;        yi[c][base + k] = ui - vr;
                        yr[c][base + k + half] = ur - vi; yi[c][base + k + half] = ui + vr;
                    } else {
                        const float wr = w128r[kw], wi = w128i[kw];
                        const float ar = fmaf(vr, wr, fmaf(-vi, wi, ur)), ai = fmaf(vr, wi, fmaf(vi, wr, ui));
                        yr[c][base + k] = ar;                    yi[c][base + k] = ai;
                        yr[c][base + k + half] = fmaf(2.0f, ur, -ar); yi[c][base + k + half] = fmaf(2.0f, ui, -ai);
                    }
                }
            }
        }
    }
    for (int k = 0; k < nbins; ++k) {                                      /* Z[c + na d] = y[c][d] */
        const int k2 = (nz - k) % nz, kk = k % nz;
        const float ar = yr[kk % na][kk / na], ai = yi[kk % na][kk / na];
        const float br = yr[k2 % na][k2 / na], bi = -yi[k2 % na][k2 / na];  /* conj(Z[N-k]) */
        const float sr = ar + br, si = ai + bi;
        const float orr = ar - br, oi = ai - bi;
        float tr, ti;
        CMUL(orr, oi, P->w2r[k], P->w2i[k], tr, ti);                        /* WH = 0.5 W2N */
        const float xr = fmaf(sr, 0.5f, ti);   /* E + (-i) T */
        const float xi = fmaf(si, 0.5f, -tr);
        pw[k] = fmaf(xr, xr, xi * xi);
    }
}

int orc_ft8_spectra(const int16_t *frame, float *s_out, int nbins)
{
    if (!tables_ready) make_tables();
    if (nbins < 1 || nbins > FT8_NH1 + 1) return -1;
    float x[FT8_NSPS];
    for (int j = 0; j < FT8_NHSYM; ++j) {
        const int16_t *d = frame + (size_t)FT8_NSTEP * j;
        for (int n = 0; n < FT8_NSPS; ++n) x[n] = (float)d[n];          /* the scale fac rides in stage 1's twiddle (spec v3) */
        spectrum_packed(&plan8, x, s_out + (size_t)j * nbins, nbins);
    }
    return 0;
}

typedef struct { float v; int idx; } keyed_t;
static int cmp_keyed(const void *a, const void *b)
{
    const keyed_t *x = (const keyed_t *)a, *y = (const keyed_t *)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);         /* ties: ascending index (total order) */
}

static const int icos7[7] = {3, 1, 4, 0, 6, 5, 2};

/* sync8's near-duplicate test on the time axis, "abs(candidate0(2,i)-candidate0(2,j)) .lt. 0.04" in single precision: candidate0(2,.) =
 * (jpeak - 0.5) * tstep with tstep = NSTEP / 12000.0 = fl(0.04).  0.04 is not a binary fraction, so for two entries EXACTLY ONE STEP apart the
 * outcome depends on how the two products round: of the 124 pairs (l, l + 1), l = -62 .. 61, 77 compare as closer than 0.04 s and 47 do not
 * (tests/test_sync_oracle.py pins the table); two steps apart never do.  The expression, in this order, is the spec: kernel and oracle use it,
 * tests/indep_sync.py restates it in numpy float32.  Whether upstream's compiler evaluates it the same way is not verifiable here (PARITY
 * UNPINNED); a double-precision reading would make NO one-step pair close. */
int orc_ft8_tdiff_close(int lag_i, int lag_j)
{
    const float tstep = FT8_NSTEP / 12000.0f;
    const float ti = ((float)lag_i - 0.5f) * tstep, tj = ((float)lag_j - 0.5f) * tstep;
    const float tdiff = fabsf(ti - tj);
    return tdiff < 0.04f;
}

int orc_ft8_sync(const int16_t *frame, int nfa_hz, int nfb_hz, float syncmin, int maxcand,
                 orc_candidate_t *out, int max_out,
                 float *red_o, int32_t *jpeak_o, float *red2_o, int32_t *jpeak2_o)
{
    return orc_ft8_sync_ordered(frame, nfa_hz, nfb_hz, syncmin, maxcand, ORC_ORDER_SYNC_DESC, out, max_out, red_o, jpeak_o, red2_o, jpeak2_o);
}

/* order: ORC_ORDER_SYNC_DESC -- the list strongest first and cut at maxcand in that order (the order of sync8.f90's commented-out "Sort by sync"
 * lines and of WSJT-X 1.x); ORC_ORDER_FREQ_ASC -- ascending frequency with the cut taken in THAT order (sync8.f90's live "Sort by frequency"
 * as recalled: indexx on the frequency column, copy while k <= maxcand); entries of one bin keep their order of discovery (the +-10 lag peak
 * before the +-62 one: upstream's indexx is not stable, so this tie rule is the builder's).  Neither is verifiable here: PARITY UNPINNED. */
int orc_ft8_sync_ordered(const int16_t *frame, int nfa_hz, int nfb_hz, float syncmin, int maxcand, int order,
                         orc_candidate_t *out, int max_out,
                         float *red_o, int32_t *jpeak_o, float *red2_o, int32_t *jpeak2_o)
{
    const float df = 12000.0f / FT8_NFFT1;              /* 3.125 */
    const float tstep = FT8_NSTEP / 12000.0f;           /* 0.04  */
    int ia = (int)lroundf((float)nfa_hz / df); if (ia < 1) ia = 1;
    int ib = (int)lroundf((float)nfb_hz / df);
    if (ib + 12 > FT8_NH1) ib = FT8_NH1 - 12;
    if (ib < ia) return 0;
    const int nbins = ib + 13;                           /* bins 0 .. ib+12 */
    float *s = (float *)malloc(sizeof(float) * (size_t)nbins * FT8_NHSYM);
    if (!s || orc_ft8_spectra(frame, s, nbins)) { free(s); return -1; }
#define S(i, m) s[(size_t)((m) - 1) * nbins + (i)]       /* 1-based symbol-step index m, bin i */
    const int nssy = 4, nfos = 2, jstrt = 12;            /* NSPS/NSTEP, NFFT1/NSPS, 0.5/tstep */
    const int iz = ib - ia + 1;
    float *red = (float *)calloc(FT8_NH1 + 1, sizeof(float));
    float *red2 = (float *)calloc(FT8_NH1 + 1, sizeof(float));
    int *jpeak = (int *)calloc(FT8_NH1 + 1, sizeof(int));
    int *jpeak2 = (int *)calloc(FT8_NH1 + 1, sizeof(int));
    for (int i = ia; i <= ib; ++i) {
        float best = 0, best2 = 0; int jb = 0, jb2 = 0; int first = 1, first2 = 1;
        for (int j = -FT8_JZ; j <= FT8_JZ; ++j) {
            float ta = 0, tb = 0, tc = 0, t0a = 0, t0b = 0, t0c = 0;
            for (int n = 0; n < 7; ++n) {
                const int m = j + jstrt + nssy * n;
                if (m >= 1 && m <= FT8_NHSYM) {
                    ta = ta + S(i + nfos * icos7[n], m);
                    float c0 = 0; for (int k = 0; k < 7; ++k) c0 = c0 + S(i + nfos * k, m);
                    t0a = t0a + c0;
                }
                {
                    const int mb = m + nssy * 36;
                    tb = tb + S(i + nfos * icos7[n], mb);
                    float c0 = 0; for (int k = 0; k < 7; ++k) c0 = c0 + S(i + nfos * k, mb);
                    t0b = t0b + c0;
                }
                if (m + nssy * 72 <= FT8_NHSYM) {
                    const int mc = m + nssy * 72;
                    tc = tc + S(i + nfos * icos7[n], mc);
                    float c0 = 0; for (int k = 0; k < 7; ++k) c0 = c0 + S(i + nfos * k, mc);
                    t0c = t0c + c0;
                }
            }
            float t = ta + tb + tc;
            float t0 = t0a + t0b + t0c;
            t0 = (t0 - t) / 6.0f;
            const float sync_abc = t / t0;
            t = tb + tc;
            t0 = t0b + t0c;
            t0 = (t0 - t) / 6.0f;
            const float sync_bc = t / t0;
            float sy = (sync_abc > sync_bc) ? sync_abc : sync_bc;           /* max() */
            if (!(sy == sy)) sy = 0.0f;                                     /* 0/0 on all-zero windows: defined as 0 here */
            if (j >= -10 && j <= 10 && (first || sy > best)) { best = sy; jb = j; first = 0; }   /* maxloc: first maximum */
            if (first2 || sy > best2) { best2 = sy; jb2 = j; first2 = 0; }
        }
        red[i] = best; jpeak[i] = jb; red2[i] = best2; jpeak2[i] = jb2;
    }
#undef S
    free(s);
    if (red_o) memcpy(red_o, red, sizeof(float) * (FT8_NH1 + 1));
    if (red2_o) memcpy(red2_o, red2, sizeof(float) * (FT8_NH1 + 1));
    if (jpeak_o) for (int i = 0; i <= FT8_NH1; ++i) jpeak_o[i] = jpeak[i];
    if (jpeak2_o) for (int i = 0; i <= FT8_NH1; ++i) jpeak2_o[i] = jpeak2[i];

    /* 40th-percentile normalisation (indexx -> ascending order; ties by bin) */
    keyed_t *ord = (keyed_t *)malloc(sizeof(keyed_t) * (size_t)iz);
    keyed_t *ord2 = (keyed_t *)malloc(sizeof(keyed_t) * (size_t)iz);
    for (int k = 0; k < iz; ++k) { ord[k].v = red[ia + k]; ord[k].idx = ia + k; ord2[k].v = red2[ia + k]; ord2[k].idx = ia + k; }
    qsort(ord, (size_t)iz, sizeof(keyed_t), cmp_keyed);
    qsort(ord2, (size_t)iz, sizeof(keyed_t), cmp_keyed);
    int npct = (int)lroundf(0.40f * (float)iz);
    int ncand = 0;
    const int maxpre = 1000;                              /* MAXPRECAND of sync8.f90 */
    orc_candidate_t *c0 = (orc_candidate_t *)calloc((size_t)(maxpre + 2), sizeof(orc_candidate_t));
    if (npct >= 1) {
        const float base = red[ord[npct - 1].idx];
        const float base2 = red2[ord2[npct - 1].idx];
        for (int i = ia; i <= ib; ++i) { red[i] = red[i] / base; red2[i] = red2[i] / base2; }
        const int lim = (maxpre < iz) ? maxpre : iz;
        for (int r = 1; r <= lim; ++r) {
            const int n = ord[iz - r].idx;                /* descending red (order fixed before normalisation) */
            if (ncand >= maxpre) break;
            if (red[n] >= syncmin && !isnan(red[n])) {
                c0[ncand].freq_bin = n; c0[ncand].time_step = jpeak[n]; c0[ncand].sync = red[n]; ncand++;
            }
            if (jpeak2[n] == jpeak[n]) continue;
            if (ncand >= maxpre) break;
            if (red2[n] >= syncmin && !isnan(red2[n])) {
                c0[ncand].freq_bin = n; c0[ncand].time_step = jpeak2[n]; c0[ncand].sync = red2[n]; ncand++;
            }
        }
    }
    /* near-dupe suppression: |df| < 4 Hz and |dt| < 0.04 s -> keep the stronger (sequential, in place) */
    for (int i = 0; i < ncand; ++i) { c0[i].freq_hz = (float)c0[i].freq_bin * df; c0[i].dt_s = ((float)c0[i].time_step - 0.5f) * tstep; }
    for (int i = 1; i < ncand; ++i) {
        for (int j = 0; j < i; ++j) {
            const float fdiff = fabsf(c0[i].freq_hz) - fabsf(c0[j].freq_hz);
            if (fabsf(fdiff) < 4.0f && orc_ft8_tdiff_close(c0[i].time_step, c0[j].time_step)) {      /* dt_s = (time_step - 0.5) * tstep, as above */
                if (c0[i].sync >= c0[j].sync) c0[j].sync = 0.0f;
                if (c0[i].sync < c0[j].sync) c0[i].sync = 0.0f;
            }
        }
    }
    /* final list, survivors only: descending sync (ties by ascending bin then lag), or ascending bin (ties by order of discovery) */
    int nout = 0;
    for (int pass = 0; pass < ncand && nout < max_out && nout < maxcand; ++pass) {
        int bi = -1;
        for (int i = 0; i < ncand; ++i) {
            if (!(c0[i].sync >= syncmin)) continue;
            if (bi < 0) { bi = i; continue; }
            const orc_candidate_t *a = &c0[i], *b = &c0[bi];
            if (order == ORC_ORDER_FREQ_ASC) { if (a->freq_bin < b->freq_bin) bi = i; continue; }
            if (a->sync > b->sync || (a->sync == b->sync && (a->freq_bin < b->freq_bin ||
                (a->freq_bin == b->freq_bin && a->time_step < b->time_step)))) bi = i;
        }
        if (bi < 0) break;
        out[nout++] = c0[bi];
        c0[bi].sync = -1.0f;
    }
    free(ord); free(ord2); free(c0); free(red); free(red2); free(jpeak); free(jpeak2);
    return nout;
}


/* ====================================================================================================
 * FT4 candidate search -- *** PARITY UNPINNED *** like everything in this file.
 * Restates, from memory of WSJT-X 2.6.x lib/ft4/getcandidates4.f90 + ft4_baseline.f90 + ft4_params.f90:
 *   NSPS=576 NFFT1=2304 NH1=1152 NSTEP=576 NMAX=72576 NHSYM=122, df = 12000/2304, Nuttall window
 *   s(i,j) = |FFT_2304(fac*dd*window)|^2 ; savg = mean_j s ; savsm = 15-bin moving average of savg
 *   baseline: savg in dB, 10 segments, lower 10 % of each -> 5-term polynomial least squares, +0.65 dB, back to power
 *   candidates: local maxima of savsm/sbase >= syncmin with parabolic interpolation, 200..4910 Hz, sorted by height.
 * Builder-defined arithmetic where upstream leaves it to libm / a fitting routine (so that the GPU can be bit-exact):
 *   log10 and 10^x are the fixed double-precision series below; the fit solves the 5x5 normal equations in the
 *   scaled variable u = (i - i0)/half_span by Gaussian elimination with partial pivoting.
 */
#define F4_NFFT   2304
#define F4_NH1    1152
#define F4_NSTEP  576
#define F4_NHSYM  122
#define F4_NA     9

static float f4_win[F4_NFFT];
static int f4_ready = 0;

static void f4_tables(void)
{
    const double pi = 3.14159265358979323846;
    if (!tables_ready) make_tables();
    for (int i = 0; i < F4_NFFT; ++i)                   /* nuttal_window */
        f4_win[i] = (float)(0.3635819 - 0.4891775 * cos(2.0 * pi * i / 2304.0) + 0.1365995 * cos(4.0 * pi * i / 2304.0)
                            - 0.0106411 * cos(6.0 * pi * i / 2304.0));
    f4_ready = 1;
}

/* s_out: NHSYM rows of 1153 floats (bins 0..1152) */
int orc_ft4_spectra(const int16_t *frame, float *s_out)
{
    if (!f4_ready) f4_tables();
    float x[F4_NFFT];
    const float fac = 1.0f / 300.0f;
    for (int j = 0; j < F4_NHSYM; ++j) {
        const int16_t *d = frame + (size_t)F4_NSTEP * j;
        for (int n = 0; n < F4_NFFT; ++n) x[n] = (fac * (float)d[n]) * f4_win[n];
        spectrum_packed(&plan4, x, s_out + (size_t)j * (F4_NH1 + 1), F4_NH1 + 1);
    }
    return 0;
}

/* fixed-arithmetic double log10 / 10^x (plain * + / only: identical on any IEEE-754 machine, CPU or GPU) */
double orc_log10_fixed(double x)
{
    if (!(x > 0.0)) return -1.0e300;
    int e = 0;
    double m = x;
    while (m >= 1.4142135623730951) { m = m * 0.5; ++e; }
    while (m < 0.7071067811865476) { m = m * 2.0; --e; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double s = 0.0;
    for (int k = 21; k >= 1; k -= 2) s = s * t2 + 1.0 / (double)k;      /* sum t^(k-1)/k, Horner in t2 */
    const double ln_m = 2.0 * t * s;
    return ((double)e * 0.6931471805599453 + ln_m) * 0.4342944819032518;
}

double orc_exp10_fixed(double y)
{
    const double z = y * 3.321928094887362;            /* y*log2(10) */
    double fl = (double)(long long)z;
    if (fl > z) fl = fl - 1.0;                         /* floor */
    const double f = (z - fl) * 0.6931471805599453;    /* in [0, ln 2) */
    double s = 1.0;
    for (int k = 22; k >= 1; --k) s = 1.0 + s * f / (double)k;          /* Taylor of exp(f), Horner */
    long long n = (long long)fl;
    double p = 1.0;
    if (n >= 0) { for (long long q = 0; q < n && q < 2000; ++q) p = p * 2.0; }
    else { for (long long q = 0; q < -n && q < 2000; ++q) p = p * 0.5; }
    return s * p;
}

static void shell_sort(float *a, int n)
{
    for (int gap = n / 2; gap > 0; gap /= 2)
        for (int i = gap; i < n; ++i) {
            const float v = a[i];
            int j = i;
            while (j >= gap && a[j - gap] > v) { a[j] = a[j - gap]; j -= gap; }
            a[j] = v;
        }
}

/* out arrays (optional): savsm_norm[1153], sbase[1153].  Returns candidate count (<= max_out). */
int orc_ft4_candidates(const int16_t *frame, float fa_hz, float fb_hz, float syncmin, int maxcand,
                       orc_candidate_t *out, int max_out, float *savsm_o, float *sbase_o)
{
    return orc_ft4_candidates_ordered(frame, fa_hz, fb_hz, syncmin, maxcand, ORC_ORDER_SYNC_DESC, out, max_out, savsm_o, sbase_o);
}

/* order: the peaks are FOUND scanning upwards and the scan stops at maxcand either way (getcandidates4.f90); ORC_ORDER_SYNC_DESC then lists them
 * by height (its indexx on the height column, read backwards), ORC_ORDER_FREQ_ASC leaves them as found.  Same entries in both. */
int orc_ft4_candidates_ordered(const int16_t *frame, float fa_hz, float fb_hz, float syncmin, int maxcand, int order,
                               orc_candidate_t *out, int max_out, float *savsm_o, float *sbase_o)
{
    const int NB1 = F4_NH1 + 1;
    float *s = (float *)malloc(sizeof(float) * (size_t)NB1 * F4_NHSYM);
    if (!s || orc_ft4_spectra(frame, s)) { free(s); return -1; }
    float savg[F4_NH1 + 1], savsm[F4_NH1 + 1], sbase[F4_NH1 + 1], sdb[F4_NH1 + 1];
    for (int i = 0; i <= F4_NH1; ++i) { savg[i] = 0.0f; savsm[i] = 0.0f; sbase[i] = 0.0f; sdb[i] = 0.0f; }
    for (int j = 0; j < F4_NHSYM; ++j)
        for (int i = 1; i <= F4_NH1; ++i) savg[i] = savg[i] + s[(size_t)j * NB1 + i];
    free(s);
    for (int i = 1; i <= F4_NH1; ++i) savg[i] = savg[i] / (float)F4_NHSYM;
    for (int i = 8; i <= F4_NH1 - 7; ++i) {
        float t = 0.0f;
        for (int q = i - 7; q <= i + 7; ++q) t = t + savg[q];
        savsm[i] = t / 15.0f;
    }
    const float df = 12000.0f / F4_NFFT;
    int nfa = (int)(fa_hz / df); if (nfa < (int)lroundf(200.0f / df)) nfa = (int)lroundf(200.0f / df);
    int nfb = (int)(fb_hz / df); if (nfb > (int)lroundf(4910.0f / df)) nfb = (int)lroundf(4910.0f / df);
    int ncand = 0;
    orc_candidate_t *c0 = (orc_candidate_t *)calloc((size_t)maxcand + 1, sizeof(orc_candidate_t));
    do {
        if (nfb - nfa < 20) break;
        /* ---- baseline ---- */
        const int ia = nfa, ib = (nfb < F4_NH1) ? nfb : F4_NH1;
        for (int i = ia; i <= ib; ++i) sdb[i] = (float)(10.0 * orc_log10_fixed((double)savg[i]));
        const int nseg = 10, npct = 10;
        const int nlen = (ib - ia + 1) / nseg, i0 = (ib - ia + 1) / 2;
        const double half = (double)(ib - ia + 1) / 2.0;
        double S[9] = {0}, Tm[5] = {0};
        int kz = 0;
        float tmp[F4_NH1 + 1];
        for (int n = 0; n < nseg; ++n) {
            const int ja = ia + n * nlen, jb = ja + nlen - 1;
            for (int q = 0; q < nlen; ++q) tmp[q] = sdb[ja + q];
            shell_sort(tmp, nlen);
            int jp = (int)lroundf(((float)nlen * 0.01f) * (float)npct);
            if (jp < 1) jp = 1;
            if (jp > nlen) jp = nlen;
            const float base = tmp[jp - 1];
            for (int i = ja; i <= jb; ++i) {
                if (sdb[i] <= base && kz < 1000) {
                    ++kz;
                    const double u = (double)(i - i0) / half, y = (double)sdb[i];
                    double p = 1.0;
                    for (int q = 0; q < 9; ++q) { S[q] = S[q] + p; if (q < 5) Tm[q] = Tm[q] + y * p; p = p * u; }
                }
            }
        }
        if (kz < 5) break;
        double A[5][6];
        for (int r = 0; r < 5; ++r) { for (int cc = 0; cc < 5; ++cc) A[r][cc] = S[r + cc]; A[r][5] = Tm[r]; }
        int singular = 0;
        for (int col = 0; col < 5; ++col) {
            int piv = col;
            for (int r = col + 1; r < 5; ++r) if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
            if (A[piv][col] == 0.0) { singular = 1; break; }
            if (piv != col) for (int cc = 0; cc < 6; ++cc) { const double t = A[col][cc]; A[col][cc] = A[piv][cc]; A[piv][cc] = t; }
            for (int r = col + 1; r < 5; ++r) {
                const double f = A[r][col] / A[col][col];
                for (int cc = col; cc < 6; ++cc) A[r][cc] = A[r][cc] - f * A[col][cc];
            }
        }
        if (singular) break;
        double a[5];
        for (int r = 4; r >= 0; --r) {
            double t = A[r][5];
            for (int cc = r + 1; cc < 5; ++cc) t = t - A[r][cc] * a[cc];
            a[r] = t / A[r][r];
        }
        int bad = 0;
        for (int i = ia; i <= ib; ++i) {
            const double u = (double)(i - i0) / half;
            const float db = (float)(a[0] + u * (a[1] + u * (a[2] + u * (a[3] + u * a[4]))) + 0.65);
            sbase[i] = (float)orc_exp10_fixed((double)db / 10.0);
            if (!(sbase[i] > 0.0f)) bad = 1;
        }
        if (bad) break;
        for (int i = nfa; i <= nfb; ++i) savsm[i] = savsm[i] / sbase[i];
        /* ---- local maxima ---- */
        const float f_offset = -1.5f * 12000.0f / 576.0f;
        for (int i = nfa + 1; i <= nfb - 1; ++i) {
            if (savsm[i] >= savsm[i - 1] && savsm[i] >= savsm[i + 1] && savsm[i] >= syncmin) {
                const float den = savsm[i - 1] - 2.0f * savsm[i] + savsm[i + 1];
                float del = 0.0f;
                if (den != 0.0f) del = 0.5f * (savsm[i - 1] - savsm[i + 1]) / den;
                const float fpeak = ((float)i + del) * df + f_offset;
                if (fpeak < 200.0f || fpeak > 4910.0f) continue;
                const float speak = savsm[i] - 0.25f * (savsm[i - 1] - savsm[i + 1]) * del;
                c0[ncand].freq_bin = i; c0[ncand].time_step = 0; c0[ncand].sync = speak;
                c0[ncand].freq_hz = fpeak; c0[ncand].dt_s = 0.0f;
                ++ncand;
                if (ncand == maxcand) break;
            }
        }
    } while (0);
    if (savsm_o) memcpy(savsm_o, savsm, sizeof(savsm));
    if (sbase_o) memcpy(sbase_o, sbase, sizeof(sbase));
    /* descending height, ties by ascending bin */
    int nout = 0;
    for (int pass = 0; pass < ncand && nout < max_out; ++pass) {
        int bi = -1;
        for (int i = 0; i < ncand; ++i) {
            if (c0[i].time_step) continue;
            if (order == ORC_ORDER_FREQ_ASC) { if (bi < 0) bi = i; continue; }          /* as found: ascending bin */
            if (bi < 0 || c0[i].sync > c0[bi].sync || (c0[i].sync == c0[bi].sync && c0[i].freq_bin < c0[bi].freq_bin)) bi = i;
        }
        if (bi < 0) break;
        c0[bi].time_step = 1;
        out[nout] = c0[bi];
        out[nout].time_step = 0;
        ++nout;
    }
    free(c0);
    return nout;
}

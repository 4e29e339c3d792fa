/*
 * sync_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * *** PARITY UNPINNED ***  The reference (CWSL_DIGI) contains NO sync / Costas code: it hands the 12 kHz
 * int16 frame to WSJT-X's jt9.exe (source/DecoderPool.hpp:634-676).  WSJT-X is a third-party program that
 * is not vendored, not version-pinned and not present in this container, so there
 * is nothing to pin this restatement against.  It restates, from memory of the published source
 * (WSJT-X 2.6.x, lib/ft8/sync8.f90 + lib/ft8/ft8_params.f90), the FT8 candidate search:
 *
 *   NSPS=1920 NFFT1=3840 NH1=1920 NSTEP=480 NMAX=180000 NHSYM=372, df=3.125 Hz, tstep=0.04 s, JZ=62
 *   s(i,j)      = |FFT_3840( dd[480(j-1) .. +1920) / 300 , zero padded )|^2 , i = 1..NH1
 *   sync2d(i,j) = max( (ta+tb+tc)/((t0a+t0b+t0c-(ta+tb+tc))/6) , (tb+tc)/((t0b+t0c-(tb+tc))/6) )
 *                 over the three Costas arrays icos7 = 3,1,4,0,6,5,2 at symbols 0/36/72
 *   red/jpeak   = per-bin max over |lag|<=10, red2/jpeak2 over |lag|<=62, each divided by its 40th percentile
 *   candidates  : bins in descending red (up to MAXPRECAND=1000 pre-candidates: the +-10 peak and, if at a
 *                 different lag, the +-62 peak of each bin), threshold syncmin, near-dupe suppression
 *                 (4 Hz, 0.04 s), sorted by sync, first maxcand kept.
 *
 * Because no external implementation can arbitrate, the ARITHMETIC is fully specified here (float32 operations
 * in a fixed order; the transform -- "spec v3" at the top of sync_oracle.c -- is a fixed factorisation
 * 3840 -> real-pack 1920 = 15 x 128 [FT4: 2304 -> 1152 = 9 x 128], NA-point DFTs (FT8: prime-factor 3 x 5 on the eight
 * live inputs; FT4: conjugate pairs with correctly-rounded fmaf chains), radix-2 DIT butterflies of three fmaf per component, host twiddles
 * from double cos/sin with exact cardinal points; everything after the spectra is un-fused + - * /)
 * so that the GPU kernels reproduce it BIT FOR BIT and "bit-identical candidate
 * lists" is a testable statement.  Ordering of the final list: descending sync, ties by ascending bin, then lag.
 */
#ifndef SYNC_ORACLE_H
#define SYNC_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FT8_NSPS   1920
#define FT8_NFFT1  3840
#define FT8_NH1    1920
#define FT8_NSTEP  480
#define FT8_NMAX   180000
#define FT8_NHSYM  372
#define FT8_JZ     62

typedef struct {
    int32_t freq_bin;
    int32_t time_step;
    float   sync;
    float   freq_hz;
    float   dt_s;
} orc_candidate_t;

/* s_out (optional): NHSYM rows of `nbins` floats, s_out[j*nbins + i] = s(i,j) for bin i in [0,nbins) */
int orc_ft8_spectra(const int16_t *frame, float *s_out, int nbins);
/* red/jpeak/red2/jpeak2 (optional, each NH1+1 entries indexed by bin) are filled for bins ia..ib BEFORE
 * normalisation.  Returns the number of candidates written (<= max_out). */
int orc_ft8_sync(const int16_t *frame, int nfa_hz, int nfb_hz, float syncmin, int maxcand,
                 orc_candidate_t *out, int max_out,
                 float *red, int32_t *jpeak, float *red2, int32_t *jpeak2);
int orc_ft8_tdiff_close(int lag_i, int lag_j);   /* the float32 near-duplicate test on the time axis (see sync_oracle.c) */
/* final order and cut of the FT8 / FT4 lists (cwslg_set_candidate_order): see sync_oracle.c */
#define ORC_ORDER_SYNC_DESC 0
#define ORC_ORDER_FREQ_ASC  1
int orc_ft8_sync_ordered(const int16_t *frame, int nfa_hz, int nfb_hz, float syncmin, int maxcand, int order,
                         orc_candidate_t *out, int max_out,
                         float *red, int32_t *jpeak, float *red2, int32_t *jpeak2);

/* ---- FT4 (getcandidates4): PARITY UNPINNED, see sync_oracle.c ---- */
#define FT4_NFFT1 2304
#define FT4_NH1   1152
#define FT4_NSTEP 576
#define FT4_NMAX  72576
#define FT4_NHSYM 122
int orc_ft4_spectra(const int16_t *frame, float *s_out /* [122][1153] */);
int orc_ft4_candidates(const int16_t *frame, float fa_hz, float fb_hz, float syncmin, int maxcand,
                       orc_candidate_t *out, int max_out, float *savsm_norm /*[1153]*/, float *sbase /*[1153]*/);
int orc_ft4_candidates_ordered(const int16_t *frame, float fa_hz, float fb_hz, float syncmin, int maxcand, int order,
                               orc_candidate_t *out, int max_out, float *savsm_norm /*[1153]*/, float *sbase /*[1153]*/);
double orc_log10_fixed(double x);
double orc_exp10_fixed(double y);

#ifdef __cplusplus
}
#endif
#endif

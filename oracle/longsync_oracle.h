/*
 * longsync_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * *** PARITY UNPINNED ***  Candidate search of the 120 s modes (BASELINE.json configs[4]).
 * CWSL_DIGI hands WSPR frames to WSJT-X's `wsprd -C <cycles> -o 5 -d <wav>` and FST4W-120 frames to
 * `jt9 -W -p 120 ... -L 1400 -H 1600 -F 200 <wav>` (source/DecoderPool.hpp:1019-1033).  WSJT-X is not vendored, not
 * version-pinned and not in this container, so -- exactly as for FT8/FT4 (sync_oracle.h) -- this file restates, from
 * memory of the published source (WSJT-X 2.6.x lib/wsprd/wsprd.c; lib/fst4/get_candidates_fst4.f90, fst4_decode.f90), the
 * part of those programs that FINDS candidates, with every arithmetic step fixed so that the GPU kernels can be
 * bit-identical to it:
 *
 * WSPR (wsprd.c main() up to and including the coarse sync search):
 *   1. readwavfile(): skips a 44-byte header.  The reference writes a 46-byte header (WaveFile.hpp:19-35), so wsprd's sample 0
 *      is the upper half of the data-length field and sample i is frame[i-1]; 114 s = 1 368 000 samples / 32768.0, zero
 *      padded to nfft1 = 1 474 560; real FFT; the 46 080 bins around i0 = 184 320 (1500 Hz) inverse-transformed -> 375 Hz
 *      complex baseband idat/qdat (/1000).
 *   2. 359 half-overlapped 512-point spectra, w[j] = sin(0.006147931 j), ps[j][i] = |FFT|^2 with the bins rotated by 256.
 *   3. psavg = sum over time; smspec = 7-bin boxcar over +-205 bins; noise = 123rd smallest of the 411; smspec/noise - 1,
 *      floored at 0.1*min_snr (min_snr = 10^-0.8); local maxima -> freq = (j-205)*df, snr = 10 log10(smspec) - 26.3;
 *      kept if |freq| <= 110 Hz; bubble-sorted by snr, descending; at most 200.
 *   4. per candidate: ifr = if0-2..if0+2, k0 = -10..21, idrift = -4..4: sync1 = ss/pow over the 162 symbols of the WSPR sync
 *      vector on sqrt(ps) at ifd-3, ifd-1, ifd+1, ifd+3; the first maximum gives (freq, shift = 128 (k0+1), drift, sync).
 *      (kindex = k0 + 2k may be negative; wsprd only tests kindex < nffts, so ps[r][kindex] then addresses the END of row
 *      r-1 of the contiguous float ps[512][nffts] -- restated as the flat index it is.)
 *
 * FST4W-120 (fst4_decode.f90 -> get_candidates_fst4.f90, iwspr = 1): see orc_fst4w_candidates below.
 *
 * Builder-defined arithmetic (nothing upstream can arbitrate): the two long transforms are evaluated as a polyphase band
 * DFT -- X[k] = sum_a W_N^(a k) Y_a[k mod M], Y_a the M-point DFT of x[R b + a] -- which is the SAME numbers as upstream's
 * zero-padded FFT followed by picking the band, at a thirtieth of the work; every M-point DFT uses "spec B" (top of
 * longsync_oracle.c: N = NA x NB, dense NA-point DFTs as four ascending fmaf chains, twiddle, NB-point radix-2 DIT
 * with (fmaf, fmaf) complex products, host tables from double cos/sin with exact cardinal points); log10 is
 * orc_log10_fixed; everything else is un-fused float + - * / sqrt in the order written.
 */
#ifndef LONGSYNC_ORACLE_H
#define LONGSYNC_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define WSPR_NPTS    1368000      /* 114 s at 12 kHz */
#define WSPR_NFFT1   1474560
#define WSPR_NFFT2   46080
#define WSPR_NDEC    32
#define WSPR_NFFTS   359
#define WSPR_MAXCAND 200

typedef struct {
    float   freq_hz;      /* relative to 1500 Hz audio */
    float   snr_db;
    float   drift;
    float   sync;
    int32_t shift;        /* 375 Hz samples */
} orc_wspr_cand_t;

/* Stage outputs are optional (NULL to skip): idat/qdat [46080], ps [512][359] row-major, smspec [411] (normalised). */
int orc_wspr_downsample(const int16_t *frame, int frame_len, float *idat, float *qdat);
int orc_wspr_search(const int16_t *frame, int frame_len, orc_wspr_cand_t *out, int max_out,
                    float *idat_o, float *qdat_o, float *ps_o, float *smspec_o);

/* spec B transform (for tests of the transform itself): n = na*nb complex points, in place, natural order in and out */
int orc_fftb(int na, int nb, float *re, float *im, int inverse);

/* ---- FST4W-120 ---- */
#define FST4W_NMAX   1440000      /* 120 s at 12 kHz = nfft1 */
#define FST4W_NSPS   8200
#define FST4W_MAXCAND 100
typedef struct {
    float   freq_hz;
    float   snr;          /* s2 peak height ("rough estimate of SNR") */
    int32_t bin;          /* index i on the df2 grid */
    int32_t pad_;
} orc_fst4w_cand_t;
/* s2_o (optional): the normalised comb spectrum, n_s2 floats indexed by i (0 outside [ina, inb]) */
int orc_fst4w_candidates(const int16_t *frame, int frame_len, int nfa_hz, int nfb_hz, float minsync,
                         orc_fst4w_cand_t *out, int max_out, float *s2_o, int n_s2, float *band_o /* optional |X|^2 of the band */);

#ifdef __cplusplus
}
#endif
#endif

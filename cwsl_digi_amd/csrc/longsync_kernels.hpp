// longsync_kernels.hpp -- candidate search of the 120 s modes on gfx950 (BASELINE.json configs[4]).
//
// *** PARITY UNPINNED by the reference ***: CWSL_DIGI hands WSPR frames to WSJT-X's `wsprd -C .. -o 5 -d` and FST4W-120 frames to
// `jt9 -W -p 120 .. -L 1400 -H 1600 -F 200` (source/DecoderPool.hpp:1019-1033); WSJT-X is not vendored.  These kernels implement
// the candidate-finding front ends of those programs (wsprd.c: FFT-based /32 down-conversion to 375 Hz, 512-point half-symbol
// spectra, smoothed-spectrum peak pick, coarse sync-vector search; get_candidates_fst4.f90: comb-summed band spectrum, 30th
// percentile, CLEAN peak pick) with the arithmetic specification of oracle/longsync_oracle.c, operation for operation, so the
// candidate lists are BIT-IDENTICAL to that restatement (tests/test_gpu_longsync.py).
//
// Both modes need a band of a >1.4 M-point real FFT.  It is evaluated as a polyphase band DFT (same numbers, 1/30 of the work):
//     X[k] = sum_{a<R} W_N^(a k) Y_a[k mod M],     Y_a = M-point DFT of x[R b + a]            (WSPR: R 32, M 46080; FST4W: R 45, M 32000)
// two real sequences share one complex M-point transform, which is "spec B" (N = NA x NB; WSPR 45 x 1024, FST4W 125 x 256):
//   fftb_stage1_kernel   dense NA-point DFTs down the columns as four ascending fmaf chains per output + twiddle W_N^(b c);
//                        a 64-column tile of the input is staged in LDS once and shared by the four waves (outputs c = wave mod 4)
//   fftb_stage2_kernel   NB / 16 threads per row, sixteen points per thread: NB-point radix-2 DIT, four levels per pass in registers
// Output convention: Z[c + NA d] = y[c][d], kept as y (row-major [NA][NB]); consumers do the index arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cwslg {

constexpr int WSPR_NPTS = 1368000, WSPR_NFFT1 = 1474560, WSPR_M = 46080, WSPR_R = 32, WSPR_NFFTS = 359, WSPR_MAXCAND = 200;
constexpr int WSPR_IQ_LEN = 46336;          // 358*128 + 512: the spectra read past 46080, where wsprd's calloc'ed buffers hold zeros
constexpr int F4W_NMAX = 1440000, F4W_NSPS = 8200, F4W_M = 32000, F4W_R = 45, F4W_MAXCAND = 100, F4W_NNW = 65600;

struct FftbTables {                          // device pointers
    const float2 *wa, *wn, *wb;              // NA | [NA][NB]: wn[c][b] = W_N^(b c) | NB/2
    const float2 *wfull;                     // [NA][NA]: wfull[c][a] = wa[(a c) mod NA], the stage-1 matrix row by row
};

struct alignas(16) LongWork {                // one per channel per launch
    const int16_t *frame;
    int frame_len;
    int pad_;
    float2 *z;                               // [T][M]   packed input, later scratch
    float2 *y;                               // [T][NA][NB]
    float2 *aux;                             // WSPR: zinv [M] | yinv [M] ; FST4W: band [nband]
    float2 *iq;                              // WSPR: [WSPR_IQ_LEN]
    float *ps, *sq;                          // WSPR: [359][512] time-major
    float *vec;                              // WSPR: smspec[411] ; FST4W: s2[F4W_NNW]
    void *cand;                              // WsprCand[200] / Fst4wCand[100]
    int *ncand;
};

struct WsprCand { float freq_hz, snr_db, drift, sync; int shift; };
struct Fst4wCand { float freq_hz, snr; int bin, pad_; };

__device__ __forceinline__ float2 lcmul(float2 v, float2 w)           // spec B's complex product
{
    return make_float2(__builtin_fmaf(v.x, w.x, -(v.y * w.y)), __builtin_fmaf(v.x, w.y, v.y * w.x));
}

// ---------------------------------------------------------------------------------------------
// spec B stage 1.  grid (NB / 64, transforms, channels), FFTB_S1_NT threads: three waves, each three groups of five outputs for NA = 45 (eight waves left
// one of them two of the nine groups and the other seven one: same-box A/B on configs[4], long-sync stage 5.05-5.08 against 5.12 ms; five waves 5.18).
#ifndef CWSLG_FFTB_S1_NT
#define CWSLG_FFTB_S1_NT 192
#endif
constexpr int FFTB_S1_NT = CWSLG_FFTB_S1_NT;   // -D overrides are for A/B builds only
template <int NA, int NB>
__global__ __launch_bounds__(FFTB_S1_NT, 4) void fftb_stage1_kernel(const LongWork *__restrict__ works, FftbTables tb, const float2 *__restrict__ wfull,
                                                          int which_in, int which_out)
{
    __shared__ float2 s_z[NA][64];
    const LongWork *w = works + blockIdx.z;
    const int t = blockIdx.y, b0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float2 *zin = (which_in == 0 ? w->z : w->aux) + (size_t)t * (NA * NB);
    float2 *yout = (which_out == 0 ? w->y : w->aux + NA * NB) + (size_t)t * (NA * NB);
    {   // the wave's share of the tile: every load issued before the first LDS write (a load-wait-store loop paid one memory
        // latency per row: 32 in a row for NA = 125)
        constexpr int NWV = FFTB_S1_NT / 64;
        constexpr int PER = (NA + NWV - 1) / NWV;
        const CWSLG_GLOBAL v2f *zg = as_global(reinterpret_cast<const v2f *>(zin)) + b0 + lane;
        float2 tmp[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int a = wv + NWV * q;
            tmp[q] = make_float2(0.f, 0.f);
            if (a < NA) { const v2f v = zg[(size_t)NB * a]; tmp[q] = make_float2(v.x, v.y); }
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int a = wv + NWV * q;
            if (a < NA) s_z[a][lane] = tmp[q];
        }
    }
    __syncthreads();
    const int b = b0 + lane;
    // Five outputs per pass over the column (one LDS read of the input feeds 20 fmaf); their twiddles W_NA^(a c) are read from
    // the row-major matrix wfull[c][a] with wave-uniform addresses, i.e. by the scalar unit into SGPR operands of the fmaf.
    // (First version: (a c) mod NA tracked in scalar registers + an LDS broadcast read per 4 fmaf -- as many SALU and LDS
    // instructions as fmaf, 4.2 ms per 128 FST4W frames against a VALU time of 0.7 ms.)  Each output's chain is still its own
    // four accumulators walked in ascending a, so the bits do not change.
    constexpr int CB = 5;
    static_assert(NA % CB == 0 && NA % 5 == 0, "NA is a multiple of the output block and of the unroll");
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    for (int cg = wvu; cg < NA / CB; cg += FFTB_S1_NT / 64) {
        const int c0 = cg * CB;
        const float2 *__restrict__ wrow = wfull + (size_t)c0 * NA;    // a __restrict__ kernel argument: provably read-only, so scalar loads
        // Round 4: the four chains of an output as TWO packed chains, (P, S) += (x.x, x.y) * wa.x and (R, Q) += (x.x, x.y) * wa.y.  The twiddle is a
        // scalar-register operand, and gfx950 issues a one-lane-wide FP32 operation with a scalar source at half rate (4.2 cycles per wave64
        // instruction against 2.35: scripts/micro/pk_issue.hip) -- v_pk_fma_f32 takes the same 4.25 cycles for two of them, with or without
        // a scalar source.  Each lane of the packed operation is a correctly rounded fmaf on the same operands in the same order: same bits.
        v2f PS[CB], RQ[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) { PS[j] = v2f{0.f, 0.f}; RQ[j] = v2f{0.f, 0.f}; }
        for (int a0 = 0; a0 < NA; a0 += 5) {
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const v2f x = *reinterpret_cast<const v2f *>(&s_z[a0 + u][lane]);
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    const float2 wa = wrow[j * NA + a0 + u];
                    PS[j] = __builtin_elementwise_fma(x, v2f{wa.x, wa.x}, PS[j]);
                    RQ[j] = __builtin_elementwise_fma(x, v2f{wa.y, wa.y}, RQ[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < CB; ++j)
            gst2(yout + (size_t)(c0 + j) * NB + b, lcmul(make_float2(PS[j].x - RQ[j].y, RQ[j].x + PS[j].y), tb.wn[(size_t)(c0 + j) * NB + b]));
    }
}

// spec B stage 1 on the matrix cores.  grid (NB / 32, transforms, channels), 64 x ceil(NA / 32) threads: every wave owns one
// 32 (outputs c) x 32 (columns b) tile.  The four chains of an output are dense products over a,
//     P = Wr Zr,  Q = Wi Zi,  R = Wi Zr,  S = Wr Zi        (W[c][a] = WA[(a c) mod NA]),
// and gfx950's f32 MFMA is bit for bit a k-ordered fmaf chain (cdna_hip_programming.md, 'FP32-input MFMA'; the FT4 stage's
// ft4_dft567_mfma_kernel relies on the same fact): v_mfma_f32_32x32x2_f32 with k0 = (w[a], z[a]), k1 = (w[a+1], z[a+1]) extends
// each chain by two terms in ascending a -- exactly the restatement's order.  NA is odd: the last step's k1 carries z = 0, which
// adds exactly nothing.  Used for FST4W's 125 x 256: the VALU form (fftb_stage1_kernel, CWSLG_LONG_VARIANT bit 0) waits on
// scalar-cache misses of its 125 KB twiddle matrix (2.05 ms per 128 frames; this kernel 1.08 ms); here the twiddle is one LDS read
// and three integer operations per lane per FOUR matrix instructions (256 cycles of matrix pipe).  WSPR's 45 x 1024 stays on the
// VALU form (0.63 against 1.14 ms: 45 rows waste 30 % of two 32-row tiles and its 16 KB matrix sits in the scalar cache).
typedef float lf32x16 __attribute__((ext_vector_type(16)));
template <int NA, int NB>
__global__ __launch_bounds__(64 * ((NA + 31) / 32)) void fftb_stage1_mfma_kernel(const LongWork *__restrict__ works, FftbTables tb,
                                                                               int which_in, int which_out)
{
    __shared__ float2 s_wa[NA];
    __shared__ __attribute__((aligned(16))) float2 s_zt[NA + 1][32];                    // the tile's inputs: every wave of the workgroup multiplies the SAME 32 columns
    const LongWork *w = works + blockIdx.z;
    const int t = blockIdx.y, b0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float2 *zin = (which_in == 0 ? w->z : w->aux) + (size_t)t * (NA * NB);
    float2 *yout = (which_out == 0 ? w->y : w->aux + NA * NB) + (size_t)t * (NA * NB);
    // Round 4: the tile goes through LDS once (16-byte loads, all issued before the first LDS write).  Each wave used to read its operand of every
    // step from memory inside the matrix-instruction loop -- the same 32 KB by every wave of the workgroup, one load latency per two steps:
    // 1.07 ms per 128 frames against 0.70 ms of matrix-pipe time.
    {
        const CWSLG_GLOBAL v4f *zg4 = reinterpret_cast<const CWSLG_GLOBAL v4f *>(as_global(reinterpret_cast<const v2f *>(zin)) + b0);
        constexpr int NT = 64 * ((NA + 31) / 32);
        constexpr int NQ = (NA + 1) * 16, PER = (NQ + NT - 1) / NT;      // 16-byte pieces of the tile (two columns each); row NA is the zero row of the odd last step
        v4f tmp[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + NT * q, a = e >> 4, p = e & 15;
            tmp[q] = v4f{0.f, 0.f, 0.f, 0.f};
            if (a < NA) tmp[q] = zg4[(size_t)(NB / 2) * a + p];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int e = tid + NT * q, a = e >> 4, p = e & 15;
            if (a <= NA) *reinterpret_cast<v4f *>(&s_zt[a][2 * p]) = tmp[q];
        }
    }
    for (int k = tid; k < NA; k += blockDim.x) s_wa[k] = tb.wa[k];
    __syncthreads();
    const int i = lane & 31, h = lane >> 5;
    const int c0 = 32 * wv;
    const int c = (c0 + i < NA) ? c0 + i : 0;              // rows past NA - 1 compute row 0 and are not stored
    const int b = b0 + i;
    lf32x16 P, Q, R, S;
#pragma unroll
    for (int v = 0; v < 16; ++v) { P[v] = 0.f; Q[v] = 0.f; R[v] = 0.f; S[v] = 0.f; }
    int idx = (h * c) % NA;                               // ((a + h) c) mod NA at a = 0
    const int step = (2 * c) % NA;
#pragma unroll 2
    for (int a = 0; a < NA; a += 2) {
        const float2 z = s_zt[a + h][i];                    // lane (i, h): B[k = h][j = i] = z[a + h][b0 + i]  (row NA holds zeros)
        const float2 tw = s_wa[idx];                        //              A[i][k = h]     = W[c0 + i][a + h]
        idx += step;
        if (idx >= NA) idx -= NA;
        P = __builtin_amdgcn_mfma_f32_32x32x2f32(tw.x, z.x, P, 0, 0, 0);
        Q = __builtin_amdgcn_mfma_f32_32x32x2f32(tw.y, z.y, Q, 0, 0, 0);
        R = __builtin_amdgcn_mfma_f32_32x32x2f32(tw.y, z.x, R, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x2f32(tw.x, z.y, S, 0, 0, 0);
    }
    // D[row][col]: col = lane & 31 (b), row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5) (c)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int cr = c0 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (cr < NA) gst2(yout + (size_t)cr * NB + b, lcmul(make_float2(P[v] - Q[v], R[v] + S[v]), tb.wn[(size_t)cr * NB + b]));
    }
}

// spec B stage 2: the NB-point radix-2 DIT of every row, in place on y.  grid (ceil(NA / rows per workgroup), transforms, channels), 256 threads.
// Round 4, second form: SIXTEEN points per thread, i.e. four radix-2 levels per pass in registers -- NB / 16 threads per row (one wave for NB = 1024,
// a quarter wave for NB = 256), 4 / 16 rows per workgroup, and no workgroup barrier behind the twiddle table's: a row never leaves its wave.
//   pass 1  levels 2..16 on positions 16 t .. 16 t + 15 of the bit-reversed order.  Position p holds input brev(p), so thread t's sixteen inputs are
//           row[brev(t) + (NB / 16) m], m = 0..15 (register brev4(m)): read straight from memory (for each m the wave's reads permute one contiguous
//           run) -- no bit-reversing scatter into LDS (which put all 64 lanes of a store on one bank pair).  Twiddles W^(k NB / len): wave-uniform reads.
//   pass 2  levels 32..256 on positions 256 blk + r + 16 q, q = 0..15 (thread = (blk, r)); level 32 * 2^s pairs q with q + 2^s, twiddle index
//           (r + 16 (q mod 2^s)) NB / len.  NB = 256 ends here: stored from registers.
//   pass 3  (NB = 1024) levels 512, 1024 on positions g + 256 q, q = 0..3, g = t + 64 i: stored from registers.
// The LDS image of a row is padded by one element per sixteen (position p at p + p / 16): every pass reads and writes with consecutive lanes on
// consecutive elements or on a stride of 17.  Every butterfly is the restatement's (t = v w by lcmul, u + t, u - t) on the same operands:
// bit-identical rows.  (First form of round 4: two levels per pass, four points per thread, one row per workgroup, five barrier-separated passes
// behind a bit-reversing scatter: 0.55 ms per launch of either size against 0.25 ms for its bytes.)
__device__ __forceinline__ void fftb_bf(float2 &u, float2 &v, float2 tw)
{
    const float2 tt = lcmul(v, tw);
    const float2 u0 = u;
    u = make_float2(u0.x + tt.x, u0.y + tt.y);
    v = make_float2(u0.x - tt.x, u0.y - tt.y);
}
template <int NA, int NB>
__global__ __launch_bounds__(256) void fftb_stage2_kernel(const LongWork *__restrict__ works, FftbTables tb, int which)
{
    constexpr int LOGB = (NB == 1024) ? 10 : 8;
    static_assert(NB == 1024 || NB == 256, "NB = 256 or 1024");
    constexpr int TPR = NB / 16;                         // threads per row
    constexpr int RPW = 256 / TPR;                       // rows per workgroup
    constexpr int PITCH = NB + NB / 16;
    __shared__ float2 s_r[RPW][PITCH];
    __shared__ float2 s_w[NB / 2];
    const LongWork *w = works + blockIdx.z;
    const int tid = threadIdx.x, t = tid % TPR;
    const int c = (int)blockIdx.x * RPW + tid / TPR;     // this thread's row
    const bool live = c < NA;
    CWSLG_GLOBAL v2f *row = reinterpret_cast<CWSLG_GLOBAL v2f *>(as_global_rw(which == 0 ? w->y : w->aux + NA * NB)) + ((size_t)blockIdx.y * NA + (live ? c : 0)) * NB;
    const CWSLG_GLOBAL v2f *wb = reinterpret_cast<const CWSLG_GLOBAL v2f *>(as_global(tb.wb));
    float2 *sr = s_r[tid / TPR];
    auto pad = [](int p) { return p + (p >> 4); };
    float2 e[16];
    {   // pass 1: this thread's sixteen inputs, issued before the twiddle table's loads are waited for
        const int bt = (int)(__brev((unsigned)t) >> (32 - (LOGB - 4)));
        v2f x[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) x[m] = live ? row[bt + TPR * m] : v2f{0.f, 0.f};
        for (int k = tid; k < NB / 2; k += 256) { const v2f q = wb[k]; s_w[k] = make_float2(q.x, q.y); }
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int j = ((m & 1) << 3) | ((m & 2) << 1) | ((m & 4) >> 1) | ((m & 8) >> 3);
            e[j] = make_float2(x[m].x, x[m].y);
        }
    }
    __syncthreads();                                     // the twiddle table
#pragma unroll
    for (int len = 2; len <= 16; len <<= 1) {
        const int h = len >> 1;
#pragma unroll
        for (int base = 0; base < 16; base += len)
#pragma unroll
            for (int k = 0; k < h; ++k) fftb_bf(e[base + k], e[base + k + h], s_w[k * (NB / len)]);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) sr[pad(16 * t + j)] = e[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                     // a row lives in ONE wave: its LDS operations execute in program order
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {   // pass 2
        const int blk = t >> 4, r = t & 15;
#pragma unroll
        for (int q = 0; q < 16; ++q) e[q] = sr[pad(256 * blk + r + 16 * q)];
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            const int len = 32 << s_;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (!(q & (1 << s_))) fftb_bf(e[q], e[q | (1 << s_)], s_w[(r + 16 * (q & ((1 << s_) - 1))) * (NB / len)]);
        }
        if (NB == 256) {
            if (live)
#pragma unroll
                for (int q = 0; q < 16; ++q) row[r + 16 * q] = v2f{e[q].x, e[q].y};
            return;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) sr[pad(256 * blk + r + 16 * q)] = e[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // pass 3 (NB = 1024): levels 512 and 1024
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = t + 64 * i;
        float2 f0 = sr[pad(g)], f1 = sr[pad(g + 256)], f2 = sr[pad(g + 512)], f3 = sr[pad(g + 768)];
        const float2 w1 = s_w[g * (NB / 512)];
        fftb_bf(f0, f1, w1);
        fftb_bf(f2, f3, w1);
        fftb_bf(f0, f2, s_w[g]);
        fftb_bf(f1, f3, s_w[g + 256]);
        if (live) {
            row[g] = v2f{f0.x, f0.y}; row[g + 256] = v2f{f1.x, f1.y}; row[g + 512] = v2f{f2.x, f2.y}; row[g + 768] = v2f{f3.x, f3.y};
        }
    }
}

// Z[k] of transform t out of the row-major stage-2 output
template <int NA, int NB>
__device__ __forceinline__ float2 fftb_at(const float2 *y, int t, int k)
{
    return y[(size_t)t * (NA * NB) + (size_t)(k % NA) * NB + k / NA];
}

// ---------------------------------------------------------------------------------------------
// WSPR.  What wsprd reads as sample n of the file the reference writes: a 46-byte header of which wsprd skips 44, so sample 0
// is the upper half of the data-length field and sample n is frame[n - 1] (longsync_oracle.c:wspr_sample).
__device__ __forceinline__ float wspr_sample(const CWSLG_GLOBAL int16_t *frame, int frame_len, int n)
{
    if (n >= WSPR_NPTS) return 0.0f;
    int v;
    if (n == 0) v = (int)(short)(((unsigned)frame_len * 2u) >> 16);
    else v = (n - 1 < frame_len) ? (int)frame[n - 1] : 0;
    return (float)v * (1.0f / 32768.0f);
}

// Eight consecutive int16 samples of a frame as floats: one 16-byte global load where the chunk lies inside the frame (frames are 256-byte
// aligned and their lengths multiples of 8), element by element at its end, zeros beyond it.
__device__ __forceinline__ void load_i16x8(const CWSLG_GLOBAL int16_t *frame, int frame_len, int i0, float (&v)[8])
{
    if (i0 + 8 <= frame_len) {
        typedef unsigned v4u __attribute__((ext_vector_type(4)));
        const v4u q = *reinterpret_cast<const CWSLG_GLOBAL v4u *>(frame + i0);
        const unsigned wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] = (float)(int)(short)(wds[e] & 0xFFFFu);
            v[2 * e + 1] = (float)((int)wds[e] >> 16);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (i0 + e < frame_len) ? (float)frame[i0 + e] : 0.0f;
    }
}

// grid (M / 256, channels): 256 consecutive b = 8192 consecutive samples, read once -- 16 bytes per lane (round 3 read two: 0.51 ms per
// 128 frames) -- and dealt out to the 16 packed sequences through LDS (index n + n/32: the stride-32 reads of the deal fall on 32
// different banks).  wsprd's sample n is frame[n - 1] (wspr_sample above), so the aligned chunk frame[N0 + 8 c ...] lands on local samples
// 8 c + 1 ... 8 c + 8; local sample 0 is the last element of the previous workgroup's range (the header word for the first).
__global__ __launch_bounds__(256) void wspr_pack_kernel(const LongWork *__restrict__ works)
{
    __shared__ float s_x[WSPR_R * 256 + 256];
    const LongWork *w = works + blockIdx.y;
    const CWSLG_GLOBAL int16_t *frame = as_global(w->frame);
    const int frame_len = w->frame_len;
    const int b0 = blockIdx.x * 256, tid = threadIdx.x;
    const int N0 = WSPR_R * b0;
    constexpr int NLOC = WSPR_R * 256;
#pragma unroll
    for (int k = 0; k < NLOC / 8 / 256; ++k) {
        const int c = tid + 256 * k;
        float v[8];
        load_i16x8(frame, frame_len, N0 + 8 * c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n = 8 * c + e + 1;                      // local sample; global sample N0 + n
            if (n < NLOC) s_x[n + (n >> 5)] = (N0 + n < WSPR_NPTS) ? v[e] * (1.0f / 32768.0f) : 0.0f;
        }
    }
    if (tid == 0) s_x[0] = wspr_sample(frame, frame_len, N0);
    __syncthreads();
    CWSLG_GLOBAL v2f *z = reinterpret_cast<CWSLG_GLOBAL v2f *>(as_global_rw(w->z));
#pragma unroll 4
    for (int p = 0; p < WSPR_R / 2; ++p) {
        const int n1 = WSPR_R * tid + p, n2 = n1 + WSPR_R / 2;
        z[(size_t)p * WSPR_M + b0 + tid] = v2f{s_x[n1 + (n1 >> 5)], s_x[n2 + (n2 >> 5)]};
    }
}

// fftin[i] = sum_{a<32} Y_a[i] T_a[i] (a ascending), stored conjugated as the input of the inverse transform.
// grid (45, channels), 256 threads: the workgroup owns the residue c = i mod 45, its threads walk d (i = c + 45 d), so Z_p[i] =
// y[p][c][d] and its mirror Z_p[M - i] = y[p][45 - c][1023 - d] are read along rows (the first version gathered both across
// rows: 32 scattered 8-byte loads per output).  T is stored in the same order: T[a][c][d].
__global__ __launch_bounds__(256) void wspr_combine_kernel(const LongWork *__restrict__ works, const float2 *__restrict__ T)
{
    const LongWork *w = works + blockIdx.y;
    const int c = blockIdx.x;
    const int cm = (c == 0) ? 0 : 45 - c;
    for (int d = threadIdx.x; d < 1024; d += 256) {
        const int dm = (c == 0) ? ((1024 - d) & 1023) : 1023 - d;
        float2 A[16], B[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            A[p] = gld2(w->y + (size_t)p * WSPR_M + (size_t)c * 1024 + d);
            B[p] = gld2(w->y + (size_t)p * WSPR_M + (size_t)cm * 1024 + dm);
        }
        const float2 *Tc = T + (size_t)c * 1024 + d;
        float fr = 0.0f, fi = 0.0f;
#pragma unroll
        for (int p = 0; p < 16; ++p) {                               // a = p: Y = (Z[i] + conj Z[M-i]) / 2
            const float2 Y = make_float2((A[p].x + B[p].x) * 0.5f, (A[p].y - B[p].y) * 0.5f);
            const float2 tt = lcmul(Y, Tc[(size_t)p * WSPR_M]);
            fr = fr + tt.x; fi = fi + tt.y;
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) {                               // a = p + 16: Y = (Z[i] - conj Z[M-i]) / 2i
            const float2 Y = make_float2((A[p].y + B[p].y) * 0.5f, (B[p].x - A[p].x) * 0.5f);
            const float2 tt = lcmul(Y, Tc[(size_t)(p + 16) * WSPR_M]);
            fr = fr + tt.x; fi = fi + tt.y;
        }
        gst2(w->aux + c + 45 * d, make_float2(fr, -fi));
    }
}

// idat/qdat = conj(inverse output) / 1000 (double division, as `fftout/1000.0`), zero tail.  grid (IQ_LEN / 256, channels)
__global__ __launch_bounds__(256) void wspr_iq_kernel(const LongWork *__restrict__ works)
{
    const LongWork *w = works + blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= WSPR_IQ_LEN) return;
    float2 o = make_float2(0.f, 0.f);
    if (i < WSPR_M) {
        const float2 zc = gld2(w->aux + WSPR_M + (size_t)(i % 45) * 1024 + i / 45);
        o = make_float2((float)((double)zc.x / 1000.0), (float)((double)(-zc.y) / 1000.0));
    }
    gst2(w->iq + i, o);
}

// 359 windowed 512-point spectra.  grid (359, channels), 256 threads.  ps/sq stored time-major [i][j].
__global__ __launch_bounds__(256) void wspr_spectra_kernel(const LongWork *__restrict__ works, const float2 *__restrict__ w512,
                                                            const float *__restrict__ win)
{
    __shared__ float2 s_r[512];
    __shared__ float2 s_w[256];
    const LongWork *w = works + blockIdx.y;
    const int i = blockIdx.x, tid = threadIdx.x;
    s_w[tid] = w512[tid];
    for (int j = tid; j < 512; j += 256) {
        const float2 v = gld2(w->iq + i * 128 + j);
        const float wj = win[j];
        s_r[__brev((unsigned)j) >> 23] = make_float2(v.x * wj, v.y * wj);
    }
    __syncthreads();
    for (int len = 2; len <= 512; len <<= 1) {
        const int half = len >> 1, step = 512 / len;
        const int k = tid & (half - 1), base = (tid - k) * 2;
        const float2 u = s_r[base + k], v = s_r[base + k + half];
        const float2 tt = lcmul(v, s_w[k * step]);
        s_r[base + k] = make_float2(u.x + tt.x, u.y + tt.y);
        s_r[base + k + half] = make_float2(u.x - tt.x, u.y - tt.y);
        __syncthreads();
    }
    for (int j = tid; j < 512; j += 256) {
        const float2 v = s_r[(j + 256) & 511];
        const float p = v.x * v.x + v.y * v.y;
        as_global_rw(w->ps)[(size_t)i * 512 + j] = p;
        as_global_rw(w->sq)[(size_t)i * 512 + j] = sqrtf(p);
    }
}

__device__ __forceinline__ double long_log10_fixed(double x)          // = orc_log10_fixed (oracle/sync_oracle.c)
{
    if (!(x > 0.0)) return -1.0e300;
    int e = 0;
    double m = x;
    while (m >= 1.4142135623730951) { m = m * 0.5; ++e; }
    while (m < 0.7071067811865476) { m = m * 2.0; --e; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double s = 0.0;
    for (int k = 21; k >= 1; k -= 2) s = s * t2 + 1.0 / (double)k;
    const double ln_m = 2.0 * t * s;
    return ((double)e * 0.6931471805599453 + ln_m) * 0.4342944819032518;
}

// psavg -> smspec -> noise percentile -> peaks -> snr -> +-110 Hz -> order by snr (stable, descending).  grid (channels), 512 threads.
__global__ __launch_bounds__(512) void wspr_peaks_kernel(const LongWork *__restrict__ works)
{
    __shared__ float s_avg[512], s_sm[416];
    __shared__ float s_noise;
    __shared__ int s_flag[416], s_pos[416];
    __shared__ float s_f[WSPR_MAXCAND], s_snr[WSPR_MAXCAND];
    __shared__ int s_n;
    const LongWork *w = works + blockIdx.x;
    const int tid = threadIdx.x;
    {
        float s = 0.0f;
        for (int i = 0; i < WSPR_NFFTS; ++i) s = s + as_global(w->ps)[(size_t)i * 512 + tid];
        s_avg[tid] = s;
    }
    __syncthreads();
    if (tid < 411) {
        float s = 0.0f;
#pragma unroll
        for (int j = -3; j <= 3; ++j) s = s + s_avg[256 - 205 + tid + j];
        s_sm[tid] = s;
    }
    __syncthreads();
    if (tid < 411) {                                     // the 123rd smallest: the value whose rank interval contains 122
        const float v = s_sm[tid];
        int less = 0, eq = 0;
        for (int x = 0; x < 411; ++x) { const float o = s_sm[x]; less += (o < v) ? 1 : 0; eq += (o == v) ? 1 : 0; }
        if (less <= 122 && 122 < less + eq) s_noise = v;
    }
    __syncthreads();
    const float noise = s_noise;
    const float min_snr = 0.15848932f;                   // (float)pow(10.0, -0.8)
    float sm = 0.0f;
    if (tid < 411) {
        sm = (float)((double)(s_sm[tid] / noise) - 1.0);
        if (sm < min_snr) sm = (float)(0.1 * (double)min_snr);
    }
    __syncthreads();
    if (tid < 411) { s_sm[tid] = sm; as_global_rw(w->vec)[tid] = sm; }
    __syncthreads();
    const float df = 0.732421875f;                       // 375/256/2
    int flag = 0;
    if (tid >= 1 && tid < 410) flag = (s_sm[tid] > s_sm[tid - 1] && s_sm[tid] > s_sm[tid + 1]) ? 1 : 0;
    if (tid < 416) s_flag[tid] = flag;
    __syncthreads();
    if (tid == 0) {                                      // 411 flags: a serial compaction is short, and keeps "npk < 200" literal
        int npk = 0;
        for (int j = 1; j < 410; ++j) { s_pos[j] = -1; if (s_flag[j] && npk < WSPR_MAXCAND) s_pos[j] = npk++; }
        s_n = npk;
    }
    __syncthreads();
    if (tid >= 1 && tid < 410 && s_pos[tid] >= 0) {
        s_f[s_pos[tid]] = (float)(tid - 205) * df;
        s_snr[s_pos[tid]] = (float)(10.0 * long_log10_fixed((double)s_sm[tid]) - (double)26.3f);
    }
    __syncthreads();
    if (tid == 0) {                                      // keep |freq| <= 110 Hz, in order
        int n2 = 0;
        for (int j = 0; j < s_n; ++j)
            if (s_f[j] >= -110.0f && s_f[j] <= 110.0f) { s_f[n2] = s_f[j]; s_snr[n2] = s_snr[j]; ++n2; }
        s_n = n2;
    }
    __syncthreads();
    const int npk = s_n;
    // wsprd's bubble sort (swap when snr[k] < snr[k+1]) is a stable descending sort: rank = #greater + #equal-and-earlier
    if (tid < npk) {
        const float v = s_snr[tid];
        int rank = 0;
        for (int x = 0; x < npk; ++x) { const float o = s_snr[x]; rank += (o > v || (o == v && x < tid)) ? 1 : 0; }
        CWSLG_GLOBAL WsprCand *c = as_global_rw(reinterpret_cast<WsprCand *>(w->cand)) + rank;
        c->freq_hz = s_f[tid]; c->snr_db = v; c->drift = 0.0f; c->sync = 0.0f; c->shift = 0;
    }
    if (tid == 0) *as_global_rw(w->ncand) = npk;
}

// Coarse (freq, shift, drift) search of one candidate.  grid (WSPR_MAXCAND, channels), 256 threads: 5 x 32 x 9 = 1440 combinations,
// each a sequential sum over the 162 sync-vector symbols; first maximum in (ifr, k0, idrift) order wins.
__global__ __launch_bounds__(256) void wspr_coarse_kernel(const LongWork *__restrict__ works, const unsigned char *__restrict__ pr3)
{
    __shared__ unsigned long long s_key[256];
    __shared__ float s_sgn[162];
    __shared__ float s_sq[WSPR_NFFTS * 20];
    const LongWork *w = works + blockIdx.y;
    const int j = blockIdx.x, tid = threadIdx.x;
    if (j >= *as_global(w->ncand)) return;
    CWSLG_GLOBAL WsprCand *cand = as_global_rw(reinterpret_cast<WsprCand *>(w->cand)) + j;
    if (tid < 162) s_sgn[tid] = (float)(2 * (int)pr3[tid] - 1);
    __syncthreads();
    const float df = 0.732421875f;
    const int if0 = (int)(cand->freq_hz / df + 256.0f);
    // every read of the search lies in spectrum rows if0 - 9 .. if0 + 7 (ifr = if0 +- 2, drift +- 2.73 bins, the four tones at
    // ifd +- 1, +- 3, and one row lower when a negative kindex wraps): keep columns [if0 - 10, if0 + 10) of sq[i][.] in LDS
    const int jb = if0 - 10;
    for (int e = tid; e < WSPR_NFFTS * 20; e += 256) {
        const int ii = e / 20, jc = e - ii * 20, jj = jb + jc;
        s_sq[e] = (jj >= 0 && jj < 512) ? as_global(w->sq)[(size_t)ii * 512 + jj] : 0.0f;
    }
    __syncthreads();
    unsigned long long best = 0ull;
    for (int combo = tid; combo < 1440; combo += 256) {
        const int idrift = combo % 9 - 4, k0 = (combo / 9) % 32 - 10, ifr = if0 - 2 + combo / 288;
        float ss = 0.0f, pw = 0.0f;
        for (int k = 0; k < 162; ++k) {
            const int ifd = (int)((double)ifr + ((double)(float)k - 81.0) / 81.0 * (double)(float)idrift / (2.0 * (double)df));
            const int kindex = k0 + 2 * k;
            if (kindex < WSPR_NFFTS) {
                // wsprd's ps[r][kindex] on the contiguous [512][359] array: a negative kindex is the end of row r - 1
                float p[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int jj = ifd - 3 + 2 * q, ii = kindex;
                    if (ii < 0) { ii += WSPR_NFFTS; jj -= 1; }
                    p[q] = s_sq[ii * 20 + (jj - jb)];
                }
                ss = ss + s_sgn[k] * ((p[1] + p[3]) - (p[0] + p[2]));
                pw = pw + p[0] + p[1] + p[2] + p[3];
            }
        }
        const float sync1 = ss / pw;
        if (sync1 > -1e30f) {                            // also false for NaN: such a combination never becomes the maximum
            unsigned u = __float_as_uint(sync1);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            const unsigned long long key = ((unsigned long long)u << 32) | (unsigned)(0xFFFF - combo);   // ties: the earlier combination
            best = key > best ? key : best;
        }
    }
    s_key[tid] = best;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off) { const unsigned long long o = s_key[tid + off]; if (o > s_key[tid]) s_key[tid] = o; }
        __syncthreads();
    }
    if (tid == 0 && s_key[0] != 0ull) {
        const unsigned long long key = s_key[0];
        const int combo = 0xFFFF - (int)(key & 0xFFFFu);
        unsigned u = (unsigned)(key >> 32);
        u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
        const int idrift = combo % 9 - 4, k0 = (combo / 9) % 32 - 10, ifr = if0 - 2 + combo / 288;
        cand->shift = 128 * (k0 + 1);
        cand->drift = (float)idrift;
        cand->freq_hz = (float)(ifr - 256) * df;
        cand->sync = __uint_as_float(u);
    }
}

// ---------------------------------------------------------------------------------------------
// FST4W-120.  z[p][b] = x[45 b + p] + i x[45 b + p + 22]; p = 22 carries a = 44 alone.  grid (M / 256, channels): 256 consecutive b =
// 11520 consecutive samples read once, dealt out through LDS (stride 45 is odd: conflict-free).
__global__ __launch_bounds__(256) void fst4w_pack_kernel(const LongWork *__restrict__ works)
{
    __shared__ float s_x[F4W_R * 256];
    const LongWork *w = works + blockIdx.y;
    const CWSLG_GLOBAL int16_t *frame = as_global(w->frame);
    const int lim = min(w->frame_len, F4W_NMAX);              // samples at and beyond either bound read as zero
    const int b0 = blockIdx.x * 256, tid = threadIdx.x;
    const int N0 = F4W_R * b0;                                // a multiple of 8: 16-byte loads (round 3 read two bytes per lane: 0.59 ms)
    static_assert((F4W_R * 256) % 8 == 0, "whole 16-byte chunks");
    for (int c = tid; c < F4W_R * 256 / 8; c += 256) {
        float v[8];
        load_i16x8(frame, lim, N0 + 8 * c, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s_x[8 * c + e] = v[e];
    }
    __syncthreads();
    CWSLG_GLOBAL v2f *z = reinterpret_cast<CWSLG_GLOBAL v2f *>(as_global_rw(w->z));
    for (int p = 0; p < 23; ++p) {
        const float x1 = s_x[F4W_R * tid + (p < 22 ? p : 44)];
        const float x2 = (p < 22) ? s_x[F4W_R * tid + p + 22] : 0.0f;
        z[(size_t)p * F4W_M + b0 + tid] = v2f{x1, x2};
    }
}

// c_bigfft(k) for k = jlo .. jlo + nband - 1.  grid (125, channels), 256 threads: the workgroup owns the residue c = k mod 125 and
// its threads walk d (k = c + 125 dd), so Y's two operands Z_p[i] = y[p][c][d] and Z_p[M - i] = y[p][125 - c][255 - d] are read
// along rows (the first version gathered 90 scattered 8-byte values per output: 4.1 ms per 128 frames).  T is stored in the
// same order: T[a][c][t], t = dd - dlo(c) < F4W_TP.
constexpr int F4W_TP = 200;
__global__ __launch_bounds__(256) void fst4w_band_kernel(const LongWork *__restrict__ works, const float2 *__restrict__ T, int jlo, int nband)
{
    const LongWork *w = works + blockIdx.y;
    const int c = blockIdx.x, t = threadIdx.x;
    const int dlo = (jlo - c + 124) / 125;                   // first dd with c + 125 dd >= jlo  (jlo > 125)
    const int k = c + 125 * (dlo + t);
    if (t >= F4W_TP || k >= jlo + nband) return;
    const int d = (dlo + t) & 255;
    const int cm = (c == 0) ? 0 : 125 - c, dm = (c == 0) ? ((256 - d) & 255) : 255 - d;
    const float2 *Tc = T + (size_t)c * F4W_TP + t;
    float fr = 0.0f, fi = 0.0f;
    for (int a = 0; a < F4W_R; ++a) {
        const int p = (a < 22) ? a : (a < 44 ? a - 22 : 22);
        const float2 A = gld2(w->y + (size_t)p * F4W_M + (size_t)c * 256 + d), B = gld2(w->y + (size_t)p * F4W_M + (size_t)cm * 256 + dm);
        const bool second = a >= 22 && a < 44;
        const float2 Y = second ? make_float2((A.y + B.y) * 0.5f, (B.x - A.x) * 0.5f) : make_float2((A.x + B.x) * 0.5f, (A.y - B.y) * 0.5f);
        const float2 tt = lcmul(Y, Tc[(size_t)a * (125 * F4W_TP)]);
        fr = fr + tt.x; fi = fi + tt.y;
    }
    gst2(w->aux + (k - jlo), make_float2(fr, fi));
}

struct Fst4wParams { int ina, inb, ia, ib, ndh, jlo, nnw; float df1, df2, minsync; };

// s(i), s2(i), percentile, CLEAN peak pick.  grid (channels), 512 threads; the band is at most a few hundred comb bins wide.
__global__ __launch_bounds__(512) void fst4w_cand_kernel(const LongWork *__restrict__ works, Fst4wParams P)
{
    constexpr int MAXW = 1024;                           // comb bins in the search window (1400..1600 Hz: 275)
    __shared__ float s_s[MAXW + 8], s_s2[MAXW + 8];
    __shared__ float s_base;
    __shared__ unsigned long long s_key[512];
    const LongWork *w = works + blockIdx.x;
    const int tid = threadIdx.x;
    const int lo = P.ina - 4, width = P.inb - P.ina + 9;  // local index l = i - lo, with 4 bins of zero margin each side
    for (int l = tid; l < MAXW + 8; l += 512) { s_s[l] = 0.0f; s_s2[l] = 0.0f; }
    __syncthreads();
    for (int i = P.ina + tid; i <= P.inb; i += 512) {
        const int j0 = (int)lroundf((float)i * P.df2 / P.df1);
        float acc = 0.0f;
        for (int j = j0 - P.ndh; j <= j0 + P.ndh; ++j) {
            const float2 v = gld2(w->aux + (j - P.jlo));
            acc = acc + v.x * v.x + v.y * v.y;
        }
        s_s[i - lo] = acc;
    }
    __syncthreads();
    int ina = P.ina, inb = P.inb;
    if (ina < 4) ina = 4;                                // max(ina, 1 + 3 hmod)
    if (inb > P.nnw - 3) inb = P.nnw - 3;
    for (int i = ina + tid; i <= inb; i += 512) s_s2[i - lo] = s_s[i - 3 - lo] + s_s[i - 1 - lo] + s_s[i + 1 - lo] + s_s[i + 3 - lo];
    __syncthreads();
    const int plo = ina + 3, npts = inb - ina + 1 - 6;
    if (npts < 1) { if (tid == 0) *as_global_rw(w->ncand) = 0; return; }
    int jp = (int)lroundf((float)npts * 0.01f * 30.0f);
    if (jp < 1) jp = 1;
    if (jp > npts) jp = npts;
    for (int e = tid; e < npts; e += 512) {
        const float v = s_s2[plo + e - lo];
        int less = 0, eq = 0;
        for (int x = 0; x < npts; ++x) { const float o = s_s2[plo + x - lo]; less += (o < v) ? 1 : 0; eq += (o == v) ? 1 : 0; }
        if (less <= jp - 1 && jp - 1 < less + eq) s_base = v;
    }
    __syncthreads();
    const float base = s_base;
    for (int l = tid; l < width; l += 512) s_s2[l] = s_s2[l] / base;
    __syncthreads();
    for (int i = tid; i < P.nnw; i += 512) {             // the normalised comb spectrum, for the parity tests
        const int l = i - lo;
        as_global_rw(w->vec)[i] = (l >= 0 && l < width) ? s_s2[l] : 0.0f;
    }
    int ia = P.ia, ib = P.ib;
    if (ia < 3) ia = 3;
    if (ib > P.nnw - 2) ib = P.nnw - 2;
    const float xdb[7] = {0.25f, 0.50f, 0.75f, 1.0f, 0.75f, 0.50f, 0.25f};
    CWSLG_GLOBAL Fst4wCand *out = as_global_rw(reinterpret_cast<Fst4wCand *>(w->cand));
    int ncand = 0;
    while (ncand < F4W_MAXCAND) {
        unsigned long long best = 0ull;
        for (int i = ia + tid; i <= ib; i += 512) {
            unsigned u = __float_as_uint(s_s2[i - lo]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            const unsigned long long key = ((unsigned long long)u << 32) | (unsigned)(0xFFFFFF - i);      // maxloc: first maximum
            best = key > best ? key : best;
        }
        s_key[tid] = best;
        __syncthreads();
        for (int off = 256; off >= 1; off >>= 1) {
            if (tid < off) { const unsigned long long o = s_key[tid + off]; if (o > s_key[tid]) s_key[tid] = o; }
            __syncthreads();
        }
        const unsigned long long key = s_key[0];
        __syncthreads();
        const int ip = 0xFFFFFF - (int)(key & 0xFFFFFFu);
        const float pval = s_s2[ip - lo];
        if (pval < P.minsync) break;
        if (tid < 7) {
            const int k = ip + 2 * (tid - 3);
            if (k >= ia && k <= ib) {
                const float v = s_s2[k - lo] - 0.9f * pval * xdb[tid];
                s_s2[k - lo] = (v > 0.0f) ? v : 0.0f;
            }
        }
        if (tid == 0) { CWSLG_GLOBAL Fst4wCand *c = out + ncand; c->freq_hz = P.df2 * (float)ip; c->snr = pval; c->bin = ip; c->pad_ = 0; }
        ++ncand;
        __syncthreads();
    }
    if (tid == 0) *as_global_rw(w->ncand) = ncand;
}

} // namespace cwslg

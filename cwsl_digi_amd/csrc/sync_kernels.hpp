// sync_kernels.hpp -- FT8 symbol spectra + Costas-array sync search on gfx950.
//
// *** PARITY UNPINNED by the reference ***: CWSL_DIGI has no sync code; it spawns WSJT-X jt9.exe
// (source/DecoderPool.hpp:634-676), which is not vendored, not version-pinned and not in the build
// container.  These kernels implement the published FT8 candidate search (WSJT-X 2.6.x lib/ft8/sync8.f90,
// ft8_params.f90: NSPS=1920, NFFT1=3840, NSTEP=480, NHSYM=372, JZ=62, icos7 = 3,1,4,0,6,5,2) with an
// arithmetic specification shared with the repository's own CPU restatement (oracle/sync_oracle.c):
// every float operation is in a fixed order (fused only where the restatement says fmaf), the FFT factorisation and
// twiddle tables are fixed ("spec v3", round 4), so the candidate lists are BIT-IDENTICAL to that restatement (tests/test_gpu_sync.py).
//
// Launches per slot boundary, all FT8 channels of the group batched in each:
//   symbol_spectra_v2_kernel one workgroup per (12 symbol steps, channel): 1920 int16 -> packed 1920-point complex FFT
//                         (15 x 128: prime-factor 15-point DFTs on the 8 live inputs + radix-2 DIT butterflies in LDS) -> |X|^2 rows
//   ft8_sync_chan_kernel  one workgroup per channel: walks the 32-bin bands of the search range with a sliding LDS window (the next
//                         band's lines in flight under the search), lane = two adjacent time lags, Costas correlation for 125 lags
//                         with the LDS traffic of a bin as one hand-scheduled stream (sync2d_asm.inc), wavefront arg-max for the +-10
//                         and +-62 lag peak searches; then the candidate selection of the channel: 40th-percentile normalisation
//                         (bitonic sort in LDS), thresholding, near-duplicate suppression, final ordering
//   (boundaries with fewer channels than two per CU: ft8_sync2d_v3_kernel, one workgroup per (band, channel), + ft8_candidates_kernel)
// Why not one fused kernel: one slot's spectra are 372 x ~973 floats = 1.45 MB, nine times the CU's LDS, and every lag touches 21
// symbol steps spread over the whole slot; the spectra make one trip through memory between the two launches instead (1.45 MB per
// slot, written once and fetched once, against 23 MB of IQ).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cwslg {

constexpr int FT8_NSPS = 1920, FT8_NSTEP = 480, FT8_NHSYM = 372, FT8_NH1 = 1920, FT8_JZ = 62, FT8_NMAX = 180000;
// FT4 (getcandidates4.f90, ft4_params.f90): Nuttall-windowed 2304-point spectra every 576 samples
constexpr int FT4_NFFT1 = 2304, FT4_NSTEP = 576, FT4_NHSYM = 122, FT4_NH1 = 1152, FT4_NMAX = 72576;
constexpr int FT4_ROW = 1168;           // stored row length (bins 0..1152, padded to a multiple of 16)
constexpr int SYNC_BAND = 32;           // bins per sync2d workgroup
constexpr int SYNC_MAXCAND_CAP = 600;   // MAXCAND of the upstream decoder
constexpr int SYNC_MAXPRE = 1000;       // MAXPRECAND of sync8.f90: pre-candidates before de-duplication

struct SyncConfig {
    bool enabled = false;
    float syncmin = 1.5f;
    float syncmin_ft4 = 1.2f;             // ft4_decode's threshold for getcandidates4
    bool ft4_coherent = true;             // run ft4_downsample + sync4d on every FT4 candidate (ft4sync_kernels.hpp)
    int max_cand = 200;
    int order = 0;                        // cwslg_set_candidate_order: 0 strongest first (cut at max_cand in that order), 1 ascending frequency (cut in THAT order)
    int f_lo_hz = 200, f_hi_hz = 3000;
    int ia = 64, ib = 960, nbins = 976;   // derived: bin range and stored row length (ib+13 rounded up to 16)
};

struct SyncTables {                       // device pointers: W_NZ, W_128, 0.5 W_2NZ twiddles, optional window
    const float2 *w1920, *w128, *w3840;
    const float *win;
};

struct SyncShared {
    float2 *d_tables = nullptr;           // FT8: w1920[1920] | w128[64] | w3840[1921] ; FT4: w1152[1152] | w2304[1153]
    float *d_win = nullptr;               // Nuttall window, 2304 floats
    SyncTables t{};                       // FT8 set
    SyncTables t4{};                      // FT4 set (w15 -> W_9, w1920 -> W_1152, w3840 -> W_2304)
    float2 *d_ft4c = nullptr;             // FT4 coherent-sync tables (ft4sync_kernels.hpp), one allocation
    float *d_ft4c_win = nullptr;
};

struct SyncChannelBuffers {
    char *d_block = nullptr;
    float *d_spectra = nullptr;           // [NHSYM][nbins]
    float *d_red = nullptr, *d_red2 = nullptr;      // [NH1+1]
    int *d_jpeak = nullptr, *d_jpeak2 = nullptr;    // [NH1+1]
    struct Cand { int freq_bin, time_step; float sync, freq_hz, dt_s; } *d_cand = nullptr;   // [max_cand]
    int *d_ncand = nullptr;
    int nbins = 0, max_cand = 0;
    bool ft4 = false;                     // FT4 layout: spectra [122][FT4_ROW]; red = normalised savsm, red2 = sbase
    // FT4 coherent sync (ft4sync_kernels.hpp): frame spectrum, its stage-A scratch, refined records
    char *d_ft4c = nullptr;
    float2 *d_y = nullptr, *d_cx = nullptr, *d_cd_dbg = nullptr;
    void *d_rec = nullptr;                // Ft4Rec [max_cand][3]
    int *d_nrec = nullptr;                // [max_cand]
};

struct alignas(16) SyncWork {
    const int16_t *frame;
    float *spectra;
    float *red, *red2;
    int *jpeak, *jpeak2;
    SyncChannelBuffers::Cand *cand;
    int *ncand;
    // Round 6: the slot's finalise (prepareAudio + int16, Instance.cpp:294-338, 238-241) done by the FT8 spectra kernel itself -- fin.frame != nullptr:
    // `frame` has NOT been written yet; every workgroup of symbol_spectra_v2_kernel first converts the samples its own windows cover (fin.out == frame).
    FinWork fin;
};

inline void sync_free_channel(SyncChannelBuffers &b)
{
    if (b.d_block) (void)hipFree(b.d_block);
    if (b.d_ft4c) (void)hipFree(b.d_ft4c);
    b = SyncChannelBuffers();
}
inline void sync_free_shared(SyncShared &s)
{
    if (s.d_tables) (void)hipFree(s.d_tables);
    if (s.d_win) (void)hipFree(s.d_win);
    if (s.d_ft4c) (void)hipFree(s.d_ft4c);
    if (s.d_ft4c_win) (void)hipFree(s.d_ft4c_win);
    s = SyncShared();
}

__device__ __forceinline__ float2 cmul_u(float2 a, float2 b)          // un-fused complex product
{
    const float ac = a.x * b.x, bd = a.y * b.y, ad = a.x * b.y, bc = a.y * b.x;
    return make_float2(ac - bd, ad + bc);
}
// the transform's complex product (twiddles behind stage 1, unpack): one rounding fewer per component, same bits as the oracle's CMUL
__device__ __forceinline__ float2 cmul_f(float2 v, float2 w)
{
    return make_float2(__builtin_fmaf(v.x, w.x, -(v.y * w.y)), __builtin_fmaf(v.x, w.y, v.y * w.x));
}

// ---------------------------------------------------------------------------------------------
// Symbol spectra.  grid (NHSYM, n_channels), 256 threads.
//
// The transform is DEFINED (oracle/sync_oracle.c, "spec v3") as: pack z[m] = x[2m] + i x[2m+1] (m < NPACK, zero
// above), NZ = NA x 128: NA-point DFTs over a (m = 128a + b) -- FT8: prime-factor 3 x 5 on the eight live inputs, FT4: conjugate
// pairs with fmaf chains -> twiddle W_NZ^(bc) -> NA radix-2 DIT FFTs of 128 points (bit-reversed input, three-fmaf butterflies) ->
// real-input unpack with 0.5 W_2NZ^k.
// Any schedule that evaluates the same operations gives the same bits; here each lane does 8-point groups (three
// radix-2 stages) in registers per LDS pass.  In the first three stages the butterflies with the exact table entries W^0 = (1,0) and
// W128^32 = (0,-1) are plain additions BY DEFINITION of the spec (from len = 16 on every butterfly takes the fmaf form, whatever its twiddle).
// LDS image of the NA x 128 work array: row pitch 144 complex (consecutive rows start 32 banks apart: the two rows of a
// 32-lane ds_read_b64 group fall into different halves of the 64 banks) and, after pass A, logical column i stored at
// i ^ ((i >> 3) & 15) (a 16-lane ds_write_b64 group -- one row, 16 eight-point groups -- hits 16 distinct bank pairs).
// Every pass is then conflict-free; stage 3 walks the bins residue by residue (k = NA q + r, lane = q) so that its two
// reads per bin run along rows, and stages the power row through LDS for a coalesced store.
// (History: the plain [15][128] image ran the LDS at 94 % busy, two thirds of it conflict cycles; pitch 129 with a
// 3-bit swizzle left every pass 2-way conflicted, 37 % of the LDS cycles.)
constexpr int SY_PITCH = 144;
__device__ __forceinline__ int sy_col(int i) { return i ^ ((i >> 3) & 15); }

// NA = 15 (FT8: 1920 = 15 x 128, only the first NPACK = 960 packed inputs are non-zero) or 9 (FT4: 1152 = 9 x 128).
// HALF 0 (waves 0-1): output 0 and the pairs c = 1..SPLIT;  HALF 1 (waves 2-3): the pairs c = SPLIT+1..NA/2.
// the W_NZ^(bc) twiddles of one thread's outputs, fetched at the top of the kernel so that their L2 latency runs under
// the frame load instead of after the first barrier
template <int NA>
struct Stage1Tw { float2 v[(NA == 15) ? 8 : 2 * (NA / 2 - (NA / 2) / 2)]; };
// NA = 15 (spec v3, prime-factor stage 1): the outputs of a lane are c = (10 k1 + 6 k2) mod 15 for ITS k2 -- HALF 0 (waves 0-1): k2 = 0, 1, 4,
// HALF 1 (waves 2-3): k2 = 2, 3 -- in the order (k2, k1); output 0 takes no twiddle
__host__ __device__ constexpr int pfa15_k2(int half, int i) { return half ? (i == 0 ? 2 : 3) : (i == 0 ? 0 : i == 1 ? 1 : 4); }
__host__ __device__ constexpr int pfa15_c(int k1, int k2) { return (10 * k1 + 6 * k2) % 15; }
template <int HALF, int NA>
__device__ __forceinline__ void stage1_load_tw(const float2 *__restrict__ wn, int b, Stage1Tw<NA> &tw)
{
    if constexpr (NA == 15) {
        int n = 0;
#pragma unroll
        for (int i = 0; i < (HALF ? 2 : 3); ++i)
#pragma unroll
            for (int k1 = 0; k1 < 3; ++k1) {
                const int c = pfa15_c(k1, pfa15_k2(HALF, i));
                if (c != 0) tw.v[n++] = wn[b * c];
            }
    } else {
        constexpr int NPAIR = NA / 2, SPLIT = NPAIR / 2;
        constexpr int C0 = HALF ? SPLIT + 1 : 1, C1 = HALF ? NPAIR : SPLIT;
#pragma unroll
        for (int c = C0; c <= C1; ++c) {
            tw.v[2 * (c - C0)] = wn[b * c];
            tw.v[2 * (c - C0) + 1] = wn[b * (NA - c)];
        }
    }
}

// W_9^k = (cos, -sin)(2 pi k / 9), k <= 4, as COMPILE-TIME constants (FT4's stage 1; FT8's prime-factor stage 1 has W_5 and sin 2 pi / 3 below): the
// values of the oracle's table -- float(cos), float(-sin) of the double angle, W^0 exact -- written out as hexadecimal floats; sync_ensure_shared
// recomputes them at start-up and refuses to run if a single bit differs.  Round 4: they used to arrive as kernel arguments, i.e. in SCALAR registers,
// and on gfx950 a one-lane-wide FP32 operation with a scalar-register source issues at HALF rate (4.2 against 2.35 cycles per wave64 instruction and
// SIMD; literal and inline constants run at full rate: scripts/micro/pk_issue.hip, profiles/r4_pk_issue_operands.txt).  As literals they are folded
// into the instruction word.  (It did not shorten the launch: the scalar-source port is shared by the SIMD's waves and a third of a mixed stream stays under it.)
template <int NA> __host__ __device__ constexpr float small_wr(int k);
template <int NA> __host__ __device__ constexpr float small_wi(int k);
template <> __host__ __device__ constexpr float small_wr<9>(int k)
{
    return k == 0 ? 1.0f : k == 1 ? 0x1.8836fap-1f : k == 2 ? 0x1.63a1a8p-3f : k == 3 ? -0x1.0p-1f : -0x1.e11f64p-1f;
}
template <> __host__ __device__ constexpr float small_wi<9>(int k)
{
    return k == 0 ? 0.0f : k == 1 ? -0x1.491b76p-1f : k == 2 ? -0x1.f838b8p-1f : k == 3 ? -0x1.bb67aep-1f : -0x1.5e3a88p-2f;
}

// W_5^k = (cos, -sin)(2 pi k / 5), k <= 2, and -sin(2 pi / 3): spec v3's prime-factor stage 1 (checked against the host's libm like W_15 / W_9)
__host__ __device__ constexpr float w5_r(int k) { return k == 0 ? 1.0f : k == 1 ? 0x1.3c6ef4p-2f : -0x1.9e377ap-1f; }
__host__ __device__ constexpr float w5_i(int k) { return k == 0 ? 0.0f : k == 1 ? -0x1.e6f0e2p-1f : -0x1.2cf23p-1f; }
__host__ __device__ constexpr float w5r_at(int m) { return m <= 2 ? w5_r(m) : w5_r(5 - m); }          // m = (n2 k2) mod 5
__host__ __device__ constexpr float w5i_at(int m) { return m <= 2 ? w5_i(m) : -w5_i(5 - m); }

constexpr float W3_S = -0x1.bb67aep-1f;                  // float(-sin(2 pi / 3))

// Spec v3 stage 1 for NA = 15: the 15-point DFT of the EIGHT live inputs of column b (z_7 = 0 in the columns that have seven) by the
// prime-factor algorithm, a = (5 n1 + 3 n2) mod 15, c = (10 k1 + 6 k2) mod 15 (oracle/sync_oracle.c, dft15_pfa8): five-point sums over the
// live n2 of each n1 in conjugate pairs (k2, 5 - k2), then three-point DFTs over n1, then the twiddle W_NZ^(bc).  A lane does this for ITS
// k2 (HALF 0: 0, 1, 4; HALF 1: 2, 3): 128 / 96 arithmetic instructions against the 146 / 176 of spec v2's pruned direct form.
template <int HALF>
__device__ __forceinline__ void stage1_pfa15(const float2 (&z)[8], float2 (*s_y)[SY_PITCH], const float2 *tw, int b)
{
    constexpr int KP = HALF ? 2 : 1;                       // the lane's conjugate pair (KP, 5 - KP)
    float2 Ya[3], Yb[3];                                   // Y[n1][KP], Y[n1][5 - KP]
    {   // n1 = 0: a = 0 (n2 = 0), 3 (n2 = 1), 6 (n2 = 2)
        constexpr float w1r = w5r_at((1 * KP) % 5), w1i = w5i_at((1 * KP) % 5), w2r = w5r_at((2 * KP) % 5), w2i = w5i_at((2 * KP) % 5);
        float P = z[3].x * w1r, Q = z[3].y * w1i, R = z[3].x * w1i, S = z[3].y * w1r;
        P = __builtin_fmaf(z[6].x, w2r, P); Q = __builtin_fmaf(z[6].y, w2i, Q); R = __builtin_fmaf(z[6].x, w2i, R); S = __builtin_fmaf(z[6].y, w2r, S);
        Ya[0] = make_float2(z[0].x + (P - Q), z[0].y + (R + S));
        Yb[0] = make_float2(z[0].x + (P + Q), z[0].y + (S - R));
    }
    {   // n1 = 1: a = 5 (n2 = 0), 2 (n2 = 4)
        constexpr float wr = w5r_at((4 * KP) % 5), wi = w5i_at((4 * KP) % 5);
        const float P = z[2].x * wr, Q = z[2].y * wi, R = z[2].x * wi, S = z[2].y * wr;
        Ya[1] = make_float2(z[5].x + (P - Q), z[5].y + (R + S));
        Yb[1] = make_float2(z[5].x + (P + Q), z[5].y + (S - R));
    }
    {   // n1 = 2: a = 1 (n2 = 2), 4 (n2 = 3), 7 (n2 = 4); no n2 = 0 input
        constexpr float w1r = w5r_at((2 * KP) % 5), w1i = w5i_at((2 * KP) % 5), w2r = w5r_at((3 * KP) % 5), w2i = w5i_at((3 * KP) % 5),
                        w3r = w5r_at((4 * KP) % 5), w3i = w5i_at((4 * KP) % 5);
        float P = z[1].x * w1r, Q = z[1].y * w1i, R = z[1].x * w1i, S = z[1].y * w1r;
        P = __builtin_fmaf(z[4].x, w2r, P); Q = __builtin_fmaf(z[4].y, w2i, Q); R = __builtin_fmaf(z[4].x, w2i, R); S = __builtin_fmaf(z[4].y, w2r, S);
        P = __builtin_fmaf(z[7].x, w3r, P); Q = __builtin_fmaf(z[7].y, w3i, Q); R = __builtin_fmaf(z[7].x, w3i, R); S = __builtin_fmaf(z[7].y, w3r, S);
        Ya[2] = make_float2(P - Q, R + S);
        Yb[2] = make_float2(P + Q, S - R);
    }
    int n = 0;
    auto three = [&](const float2 y0, const float2 y1, const float2 y2, int k2) {        // three-point DFT over n1 -> outputs c(k1, k2), twiddled
        const float2 x0 = make_float2((y0.x + y1.x) + y2.x, (y0.y + y1.y) + y2.y);
        const float2 t = make_float2(y1.x + y2.x, y1.y + y2.y), d = make_float2(y1.x - y2.x, y1.y - y2.y);
        const float2 m = make_float2(__builtin_fmaf(t.x, -0.5f, y0.x), __builtin_fmaf(t.y, -0.5f, y0.y));
        const float2 x1 = make_float2(__builtin_fmaf(-W3_S, d.y, m.x), __builtin_fmaf(W3_S, d.x, m.y));
        const float2 x2 = make_float2(__builtin_fmaf(W3_S, d.y, m.x), __builtin_fmaf(-W3_S, d.x, m.y));
        const int c0 = pfa15_c(0, k2), c1 = pfa15_c(1, k2), c2 = pfa15_c(2, k2);
        if (c0 == 0) s_y[0][b] = make_float2(x0.x * (1.0f / 300.0f), x0.y * (1.0f / 300.0f));       // WN'^0 = (fac, 0): plain products
        else s_y[c0][b] = cmul_f(x0, tw[n++]);
        s_y[c1][b] = cmul_f(x1, tw[n++]);
        s_y[c2][b] = cmul_f(x2, tw[n++]);
    };
    if (HALF == 0) {
        const float2 d0 = make_float2((z[0].x + z[3].x) + z[6].x, (z[0].y + z[3].y) + z[6].y);       // the k2 = 0 sums, sequential in ascending n2
        const float2 d1 = make_float2(z[5].x + z[2].x, z[5].y + z[2].y);
        const float2 d2 = make_float2((z[1].x + z[4].x) + z[7].x, (z[1].y + z[4].y) + z[7].y);
        three(d0, d1, d2, 0);
    }
    three(Ya[0], Ya[1], Ya[2], KP);
    three(Yb[0], Yb[1], Yb[2], 5 - KP);
}

template <int HALF, int NA, int AMAX>
__device__ __forceinline__ void spectra_stage1_regs(const float2 (&z)[AMAX], float2 (*s_y)[SY_PITCH], const Stage1Tw<NA> &twp, int b);

template <int HALF, int NA, int NPACK>
__device__ __forceinline__ void spectra_stage1(const float *s_x, float2 (*s_y)[SY_PITCH], const Stage1Tw<NA> &twp, int b)
{
    constexpr int AMAX = (NPACK + 127) / 128;            // 8 (FT8), 9 (FT4)
    float2 z[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        const int m = 128 * a + b;                       // a zero input adds exactly nothing to an fmaf chain
        z[a] = (m < NPACK) ? make_float2(s_x[2 * m], s_x[2 * m + 1]) : make_float2(0.f, 0.f);
    }
    spectra_stage1_regs<HALF, NA, AMAX>(z, s_y, twp, b);
}

// radix-2 DIT butterfly on two registers: (u, v) -> (u + w v, u - w v), spec v3: three fmaf-class operations per component (the
// difference is 2 u - (u + w v)), six per butterfly against spec v2's eight (product, then sum and difference)
__device__ __forceinline__ void bfly(float2 &u, float2 &v, float2 w)
{
    const float ar = __builtin_fmaf(v.x, w.x, __builtin_fmaf(-v.y, w.y, u.x));
    const float ai = __builtin_fmaf(v.x, w.y, __builtin_fmaf(v.y, w.x, u.y));
    const float dr = __builtin_fmaf(2.0f, u.x, -ar);
    const float di = __builtin_fmaf(2.0f, u.y, -ai);
    u = make_float2(ar, ai);
    v = make_float2(dr, di);
}
// |X[K]|^2 from Z[K] = A and Z[NZ - K] = B (spec v3 stage 3; wh = 0.5 W_2NZ^K, the table is stored scaled)
__device__ __forceinline__ float unpack_power(float2 A, float2 B, float2 wh)
{
    B.y = -B.y;
    const float sr = A.x + B.x, si = A.y + B.y;
    const float2 o = make_float2(A.x - B.x, A.y - B.y);
    const float2 t = cmul_f(o, wh);
    const float xr = __builtin_fmaf(sr, 0.5f, t.y), xi = __builtin_fmaf(si, 0.5f, -t.x);
    return __builtin_fmaf(xr, xr, xi * xi);
}

__device__ __forceinline__ void bfly_one(float2 &u, float2 &v)        // w = W^0 = (1, 0)
{
    const float2 a = make_float2(u.x + v.x, u.y + v.y);
    const float2 d = make_float2(u.x - v.x, u.y - v.y);
    u = a;
    v = d;
}
__device__ __forceinline__ void bfly_mj(float2 &u, float2 &v)         // w = W128^32 = (0, -1): w v = (v.y, -v.x)
{
    const float2 a = make_float2(u.x + v.y, u.y - v.x);
    const float2 d = make_float2(u.x - v.y, u.y + v.x);
    u = a;
    v = d;
}

// NA rows x 128: FT8 <15, 1920, 480, false>, FT4 <9, 2304, 576, true>.  grid (symbol steps, channels), 256 threads.
template <int NA, int NIN, int STEP, bool WINDOW>
__global__ __launch_bounds__(256) void symbol_spectra_kernel(const SyncWork *__restrict__ works, SyncTables tb, int nbins)
{
    constexpr int NZ = NA * 128;                          // packed complex length; real transform length 2*NZ
    constexpr int NPACK = NIN / 2;                        // non-zero packed inputs
    constexpr int NGRP = NA * 16;                         // 8-point groups per pass
    static_assert(NGRP <= 256 && NIN % 8 == 0, "geometry");
    __shared__ float s_x[NIN];
    __shared__ float2 s_y[NA][SY_PITCH];
    __shared__ float2 s_w128[64];
    const SyncWork *w = works + blockIdx.y;
    const int j = blockIdx.x;
    const int tid = threadIdx.x;
    const int16_t *d = w->frame + (size_t)STEP * j;
    const float fac = 1.0f / 300.0f;
    // twiddles first (tables live in L2): stage 1's W_NZ^(bc) and the W_2NZ^k of this thread's first stage-3 bins
    Stage1Tw<NA> tw1;
    if (tid < 128) stage1_load_tw<0, NA>(tb.w1920, tid & 127, tw1);
    else stage1_load_tw<1, NA>(tb.w1920, tid & 127, tw1);
    constexpr int W3PRE = 4;
    float2 w3[W3PRE];
    const bool staged = nbins <= NIN;
    const int nfull = min(nbins, NZ) / (NA * 64);         // whole 64-column chunks of bins below nbins (stage 3)
#pragma unroll
    for (int i = 0; i < W3PRE; ++i) {
        const int it = (tid >> 6) + 4 * i;
        const int k = NA * (64 * (it / NA) + (tid & 63)) + it % NA;
        w3[i] = (staged && it < NA * nfull) ? tb.w3840[k] : make_float2(0.f, 0.f);
    }
    for (int t = tid; t < NIN / 8; t += 256) {            // 16 B = 8 samples per lane
        const uint4 q = reinterpret_cast<const uint4 *>(d)[t];
        const unsigned v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float lo = (float)(short)(v[k] & 0xFFFFu), hi = (float)(short)(v[k] >> 16);
            if (NA != 15) { lo = fac * lo; hi = fac * hi; }            // FT8 (spec v3): the scale rides in stage 1's twiddle
            if (WINDOW) { lo = lo * tb.win[8 * t + 2 * k]; hi = hi * tb.win[8 * t + 2 * k + 1]; }
            s_x[8 * t + 2 * k] = lo;
            s_x[8 * t + 2 * k + 1] = hi;
        }
    }
    if (tid >= 64 && tid < 128) s_w128[tid - 64] = tb.w128[tid - 64];
    __syncthreads();

    // stage 1 (wave-uniform split of the conjugate pairs of outputs between waves 0-1 and waves 2-3)
    if (tid < 128) spectra_stage1<0, NA, NPACK>(s_x, s_y, tw1, tid & 127);
    else spectra_stage1<1, NA, NPACK>(s_x, s_y, tw1, tid & 127);
    __syncthreads();

    // stage 2, pass A: DIT stages len = 2,4,8 on logical points 8g..8g+7 of row c; the DIT input order is
    // bit-reversed, i.e. logical point i is stage-1 column bitrev7(i) = 16*bitrev3(i & 7) + bitrev4(g)
    {
        const int c = tid >> 4, g = tid & 15;
        const int gb = (int)(__brev((unsigned)g) >> 28);
        float2 e[8];
        if (tid < NGRP) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int k3 = ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1);
                e[k] = s_y[c][16 * k3 + gb];
            }
        }
        __syncthreads();                     // everyone has gathered: the image may now be rewritten
        if (tid < NGRP) {
            bfly_one(e[0], e[1]); bfly_one(e[2], e[3]); bfly_one(e[4], e[5]); bfly_one(e[6], e[7]);
            bfly_one(e[0], e[2]); bfly_mj(e[1], e[3]); bfly_one(e[4], e[6]); bfly_mj(e[5], e[7]);
            bfly_one(e[0], e[4]); bfly(e[1], e[5], s_w128[16]); bfly_mj(e[2], e[6]); bfly(e[3], e[7], s_w128[48]);
#pragma unroll
            for (int k = 0; k < 8; ++k) s_y[c][sy_col(8 * g + k)] = e[k];
        }
    }
    __syncthreads();
    // pass B: stages len = 16,32,64 on logical points {64 blk + r + 8 q : q<8}; groups: NA x 2 blocks x 8 r
    if (tid < NGRP) {
        const int c = tid >> 4, g = tid & 15;
        const int blk = g >> 3, r = g & 7;
        float2 e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = s_y[c][sy_col(64 * blk + r + 8 * q)];
        {
            const float2 w0 = s_w128[r * 8];
            bfly(e[0], e[1], w0); bfly(e[2], e[3], w0); bfly(e[4], e[5], w0); bfly(e[6], e[7], w0);
        }
        {
            const float2 w0 = s_w128[r * 4], w1 = s_w128[(r + 8) * 4];
            bfly(e[0], e[2], w0); bfly(e[1], e[3], w1); bfly(e[4], e[6], w0); bfly(e[5], e[7], w1);
        }
        {
            bfly(e[0], e[4], s_w128[r * 2]); bfly(e[1], e[5], s_w128[(r + 8) * 2]);
            bfly(e[2], e[6], s_w128[(r + 16) * 2]); bfly(e[3], e[7], s_w128[(r + 24) * 2]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) s_y[c][sy_col(64 * blk + r + 8 * q)] = e[q];
    }
    __syncthreads();
    // pass C: stage len = 128: pairs (k, k+64), twiddle W128^k ; NA*64 butterflies
    for (int idx = tid; idx < NA * 64; idx += 256) {
        const int c = idx >> 6, k = idx & 63;
        float2 u = s_y[c][sy_col(k)], v = s_y[c][sy_col(k + 64)];
        bfly(u, v, s_w128[k]);
        s_y[c][sy_col(k)] = u;
        s_y[c][sy_col(k + 64)] = v;
    }
    __syncthreads();

    // stage 3: unpack the real-input transform, power spectrum.  X[k] needs Z[k] and Z[NZ-k], Z[c + NA d] = y[c][d].
    float *out = w->spectra + (size_t)j * nbins;
    auto power_at = [&](int k) -> float {                  // generic indexing (any k <= NZ)
        const int k2 = (NZ - k) % NZ, kk = k % NZ;
        return unpack_power(s_y[kk % NA][sy_col(kk / NA)], s_y[k2 % NA][sy_col(k2 / NA)], tb.w3840[k]);
    };
    if (staged) {
        // residue-major walk: item (r, chunk) = bins k = NA (64 chunk + lane) + r.  Row k % NA = r and row (NZ-k) % NA
        // are wave-uniform, the columns run with the lane: both reads are conflict-free.  The row of powers is
        // collected in s_x (free since stage 1; stride NA between lanes = odd: conflict-free) and stored coalesced.
        float *s_pw = s_x;
        const int lane = tid & 63, wv = tid >> 6;
        int pre = 0;
        for (int it = wv; it < NA * nfull; it += 4, ++pre) {
            const int r = it % NA, q = 64 * (it / NA) + lane;
            const int k = NA * q + r;
            const float2 A = s_y[r][sy_col(q)];
            const int r2 = (r == 0) ? 0 : NA - r;
            const int q2 = (r == 0) ? ((q == 0) ? 0 : 128 - q) : 127 - q;     // (NZ - k) / NA ; k = 0 pairs with itself
            const float2 B = s_y[r2][sy_col(q2)];
            float2 wk;
            switch (pre) {                                   // the first W3PRE twiddles were fetched at the top
            case 0: wk = w3[0]; break;
            case 1: wk = w3[1]; break;
            case 2: wk = w3[2]; break;
            case 3: wk = w3[3]; break;
            default: wk = tb.w3840[k]; break;
            }
            s_pw[k] = unpack_power(A, B, wk);
        }
        for (int k = NA * 64 * nfull + tid; k < nbins; k += 256)              // ragged tail (FT8: 16 bins, FT4: bin 1152 + pad)
            s_pw[k] = (k <= NZ) ? power_at(k) : 0.0f;
        __syncthreads();
        for (int k = tid; k < nbins; k += 256) out[k] = s_pw[k];
    } else {
        for (int k = tid; k < nbins; k += 256) out[k] = (k <= NZ) ? power_at(k) : 0.0f;
    }
}

// ---------------------------------------------------------------------------------------------
// symbol_spectra_v2_kernel: the same transform (same operations, same order: bit-identical rows), two LDS round trips
// shorter.  Round-2 counters of the first version at 4096 slots: VALU 65 % and LDS 50 % busy, 807 VALU instructions per
// wave, 1236 LDS-array cycles per transform of which ~70 % are STORES (a ds_write_b64 costs 4-6 cycles against a read's
// 2).  So:
//   * the int16 window goes straight from global memory into stage 1's registers (lane = column b reads its eight
//     packed inputs 128a + b as eight coalesced 4-byte loads): no s_x image, no barrier in front of stage 1;
//   * the last radix-2 stage (len = 128) is not written back: one thread takes the butterfly PAIR
//       P1 = (row r, k = q)   and   P2 = (row NA - r, k = 63 - q)        [row 0: P2 = (0, 64 - q)]
//     whose four results are exactly the operands of the real-input unpack of four bins
//       K = NA q + r          A = u1', B = conj v2'        K = NA (63 - q) + NA - r      A = u2', B = conj v1'
//       K = NA (q + 64) + r   A = v1', B = conj u2'        K = NA (127 - q) + NA - r     A = v2', B = conj u1'
//     (Z[c + NA d] = y[c][d], X[K] pairs Z[K] with Z[NZ - K]); bins >= nbins are skipped.  That removes 15 KB of stores and
//     15 KB of loads per transform, the loop over residues and one barrier.
// Needs nbins <= NIN + 32 (the power row is staged in LDS).
template <int HALF, int NA, int AMAX>
__device__ __forceinline__ void spectra_stage1_regs(const float2 (&z)[AMAX], float2 (*s_y)[SY_PITCH], const Stage1Tw<NA> &twp, int b)
{
    if constexpr (NA == 15) {                            // spec v3: prime-factor 3 x 5 on the eight live inputs
        static_assert(AMAX == 8, "FT8: 960 packed inputs = 8 rows of 128");
        stage1_pfa15<HALF>(z, s_y, twp.v, b);
    } else {                                             // FT4 (NA = 9): conjugate pairs of fmaf chains
    constexpr int NPAIR = NA / 2, SPLIT = NPAIR / 2;
    constexpr int C0 = HALF ? SPLIT + 1 : 1, C1 = HALF ? NPAIR : SPLIT;
    const float2 *tw = twp.v;
    if (HALF == 0) {
        float2 s0 = z[0];
#pragma unroll
        for (int a = 1; a < AMAX; ++a) { s0.x = s0.x + z[a].x; s0.y = s0.y + z[a].y; }
        s_y[0][b] = s0;                                  // W_NZ^0 = 1: no multiply
    }
    // the chains start from a +0 the compiler cannot see through: fmaf(z, W, 0) would be a VOP3 v_fma_f32, which takes no literal --
    // the constant would go back into a scalar register (half rate); with a register addend it is v_fmamk_f32 (VOP2 + literal)
    float zero = 0.0f;
    asm volatile("" : "+v"(zero));
#pragma unroll
    for (int c = C0; c <= C1; ++c) {
        float P = zero, Q = zero, R = zero, S = zero;
#pragma unroll
        for (int a = 1; a < AMAX; ++a) {
            const int idx = (a * c) % NA;
            const float wr = small_wr<NA>((idx <= NA / 2) ? idx : NA - idx);
            const float wi = (idx <= NA / 2) ? small_wi<NA>(idx) : -small_wi<NA>(NA - idx);
            P = __builtin_fmaf(z[a].x, wr, P);
            Q = __builtin_fmaf(z[a].y, wi, Q);
            R = __builtin_fmaf(z[a].x, wi, R);
            S = __builtin_fmaf(z[a].y, wr, S);
        }
        const float2 yc = make_float2(z[0].x + (P - Q), z[0].y + (R + S));
        const float2 yn = make_float2(z[0].x + (P + Q), z[0].y + (S - R));
        s_y[c][b] = cmul_f(yc, tw[2 * (c - C0)]);
        s_y[NA - c][b] = cmul_f(yn, tw[2 * (c - C0) + 1]);
    }
    }
}

#ifndef CWSLG_SPEC_WAVES
#define CWSLG_SPEC_WAVES 4
#endif
#ifndef CWSLG_SPEC_HOISTPTR
#define CWSLG_SPEC_HOISTPTR 1
#endif
#ifndef CWSLG_SPEC_TIGHT
#define CWSLG_SPEC_TIGHT 1             // 0: the addressing / prefetch forms of round 4 (the A/B partner: scripts/gpu_r5_dppmax.sh with SWITCH=CWSLG_SPEC_TIGHT)
#endif
// The spectra plane (6 GB per 4096-slot boundary) is written once by symbol_spectra_v2_kernel and read once by the search: CWSLG_PLANE_NT selects
// non-temporal stores (bit 0) / loads (bit 1) for it (round 5 A/B: scripts/gpu_r5_plane_nt.sh).
#ifndef CWSLG_PLANE_NT
#define CWSLG_PLANE_NT 0
#endif
__device__ __forceinline__ void plane_store(CWSLG_GLOBAL v4f *p, v4f v)
{
    if (CWSLG_PLANE_NT & 1) __builtin_nontemporal_store(v, p); else *p = v;
}
__device__ __forceinline__ v4f plane_load4(const CWSLG_GLOBAL v4f *p)
{
    return (CWSLG_PLANE_NT & 2) ? __builtin_nontemporal_load(p) : *p;
}
__device__ __forceinline__ float plane_load1(const CWSLG_GLOBAL float *p)
{
    return (CWSLG_PLANE_NT & 2) ? __builtin_nontemporal_load(p) : *p;
}
// Symbol steps per workgroup of symbol_spectra_v2_kernel (a kernel argument).  A workgroup's prologue -- twiddles, 24 LDS addresses, the first window's
// memory latency -- is paid once per `jper` transforms: same-box at 4096 slots 4.31 ms with 12 steps (31 workgroups per channel), 4.20 with 31, 4.18 with 62;
// few channels need the short form to fill the chip (four workgroups per CU).
inline int spectra_jper(int nsteps, size_t channels)
{
    for (int jper : {62, 31}) if (channels * (size_t)((nsteps + jper - 1) / jper) >= 3072) return jper;
    return 12;
}
// (Rounds 2-3 carried a matrix-core form of spec v2's stage 1 -- its fmaf chains as v_mfma_f32_32x32x2_f32 steps -- as a lab variant: measured
// slower twice, 1.56 against 1.29 ms per 512 slots and 6.3 against 5.2 ms per 4096; it went with spec v2's stage 1.)
#if defined(CWSLG_STAMP) && defined(CWSLG_STAMP_SPEC)
// diagnostic build only (scripts/gpu_stamps_spectra.py): s_memtime of every wave at the phase seams of the workgroup's MIDDLE transform
#define PSTAMP(slot)                                                                                                               \
    do {                                                                                                                           \
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                                                  \
        if (j == j0 + (jper >> 1) && (tid_ & 63) == 0 && wg_ < 16384) {                                                                     \
            unsigned long long t_;                                                                                                 \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                                             \
            g_stamps[32 * wg_ + 8 * (tid_ >> 6) + (slot)] = t_;                                                                    \
        }                                                                                                                          \
    } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#endif
// Round 6: the slot's finalise fused into the spectra kernel (FT8 channels with the sync stage on).  finalize_kernel was a separate memory pass
// -- 4.4 GB per 4096-slot boundary, 0.87 ms -- whose int16 output this kernel then read back from HBM.  Here the workgroup converts the samples its
// own windows cover itself, writes them as the int16 frame and reads its windows back through the cache:
//   * before its first transform, the first three windows ([STEP j0, STEP (j0 + 2) + NIN), widened to whole 128-byte lines) and -- rarely, see
//     FinWork::tail_end -- its share of the frame's tail beyond the last window;
//   * then, at the top of every transform j, ONE pair of samples per lane: the next 448 / 512 samples, up to the (line-rounded) end of window j + 3.
//     The transform loop is bound by each wave's dependent chains (VALU issue 56 % busy at four waves per SIMD), so these dozen independent
//     instructions, one 8-byte load and one 4-byte store per lane ride in its idle issue slots; as a prologue over the whole span (this round's
//     first form) the conversion cost 18 us per workgroup life of 207 us -- +0.43 ms of the 0.87 ms it saved (profiles/r6_sync_ab.txt).
// The NIN - STEP samples by which a workgroup's last windows reach into its successor's share are converted by both, to the same bits: no
// workgroup waits for another.  Arithmetic per sample = finalize_kernel's, in its order: factor = 32767 / (peak + 1) * scale;
// (int16)(x * factor + 0.5f); zeros at and beyond n_valid.
// Visibility.  A store issued at the top of transform j has reached the L2 when its wave passes the top of j + 1 (the wait for that transform's
// window; the vector-memory counter retires in order), every wave of the workgroup has passed it before anyone gets through j + 1's barriers, and
// the window loaded at the top of j + 2 is window j + 3: final up to the end of its last 128-byte line (the rounding above) -- so no line is ever
// brought into the CU's L1 while a part of it is still to be written by this workgroup or holds the previous slot's samples.
struct SpectraFin {
    const CWSLG_GLOBAL float *frame;
    CWSLG_GLOBAL int16_t *out;
    unsigned nv, s_end;
    float factor;
};
__device__ __forceinline__ unsigned fin_pack2(float x0, float x1, float factor)
{
    const float s0 = x0 * factor, s1 = x1 * factor;          // buf[k] *= factor
    const float b0 = s0 + 0.5f, b1 = s1 + 0.5f;              // + 0.5f
    const int q0 = (int)b0, q1 = (int)b1;                    // C truncation toward zero, then narrowed to int16
    return ((unsigned)q0 & 0xFFFFu) | ((unsigned)q1 << 16);
}
// the range [s0, s1) (multiples of 8 samples) by all 256 lanes, 8 samples per lane and round; every load of a round before its first conversion
__device__ __forceinline__ void spectra_finalize_range(const SpectraFin &F, unsigned s0, unsigned s1)
{
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    constexpr int NC = 4;
    for (unsigned base = s0 + threadIdx.x * 8u; base < s1; base += 256u * 8u * (unsigned)NC) {
        v4f a[NC], b[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned i0 = base + (unsigned)c * (256u * 8u);
            if (i0 + 8 <= F.nv && i0 < s1) {
                a[c] = *reinterpret_cast<const CWSLG_GLOBAL v4f *>(F.frame + i0);
                b[c] = *reinterpret_cast<const CWSLG_GLOBAL v4f *>(F.frame + i0 + 4);
            } else {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (i0 + k < F.nv && i0 < s1) ? F.frame[i0 + k] : 0.0f;
                a[c] = v4f{v[0], v[1], v[2], v[3]};
                b[c] = v4f{v[4], v[5], v[6], v[7]};
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned i0 = base + (unsigned)c * (256u * 8u);
            if (i0 >= s1) break;
            v4u pk;
            pk.x = fin_pack2(a[c].x, a[c].y, F.factor); pk.y = fin_pack2(a[c].z, a[c].w, F.factor);
            pk.z = fin_pack2(b[c].x, b[c].y, F.factor); pk.w = fin_pack2(b[c].z, b[c].w, F.factor);
            *reinterpret_cast<CWSLG_GLOBAL v4u *>(F.out + i0) = pk;
        }
    }
}

// FMODE: 0 = the int16 frame exists (FT4; FT8 behind a separate finalize_kernel: lab / A-B builds); 1 = fused finalise, windows read back from the int16
// frame through the cache (this round's first product form); 2 = fused finalise, windows from an LDS ring (see below).
template <int NA, int NIN, int STEP, bool WINDOW, int FMODE = 0>
__global__ __launch_bounds__(256, CWSLG_SPEC_WAVES) void symbol_spectra_v2_kernel(const SyncWork *__restrict__ works, SyncTables tb, int nbins, int nsteps, int jper)
{
    static_assert(FMODE == 0 || (!WINDOW && NIN == 4 * STEP && STEP == 480), "the fused forms are FT8's");
    constexpr int NZ = NA * 128;
    constexpr int NPACK = NIN / 2;
    constexpr int NGRP = NA * 16;
    constexpr int AMAX = (NPACK + 127) / 128;             // 8 (FT8), 9 (FT4)
    constexpr int NH = NA / 2;                            // row pairs (r, NA - r), r = 1..NH; row 0 pairs with itself
    constexpr int NITEM = (NH + 1) * 64, IPT = (NITEM + 255) / 256;
    static_assert(NGRP <= 256, "geometry");
    __shared__ __attribute__((aligned(16))) float s_pw[NIN + 32];     // one power row: nbins <= NIN + 32 (FT8: the widest search stores 1952)
    __shared__ float2 s_y[NA][SY_PITCH];
    __shared__ float2 s_w128[64];
    // FMODE 2 (measured alternative, lab library): the int16 window as an LDS ring of sample PAIRS (one dword = two int16 samples, the unit stage 1 reads):
    // RING = a window (960 pairs) + one step (240 pairs), every pair stored twice, RING apart, so that a lane's eight reads base + 128 a need no wrap.  The
    // workgroup converts 480 samples per transform (lanes 0..239, one pair each) into the ring AND into the int16 frame in HBM -- which the kernel then never
    // reads.  The idea: the frame's stores cost 0.5 ms per 1.5 GB when the same CU reads them back two transforms later, the plane's write-only 6 GB cost
    // 0.17 ms.  The result: 5.1-5.2 against 4.5 ms (profiles/r6_sync_ab.txt), with the ring's reads at the head of the transform or prefetched into registers behind the
    // previous transform's first barrier alike: ten more LDS operations per lane and transform cost this kernel more than the read-back they avoid.
    constexpr int RING = NPACK + STEP / 2;                 // 1200 pairs
    __shared__ unsigned s_ring[FMODE == 2 ? 2 * RING : 1];
    const SyncWork *w = works + blockIdx.y;
    const int j0 = blockIdx.x * jper;
    const int jend = min(j0 + jper, nsteps);
    const int tid_ = threadIdx.x;
    const int b_ = tid_ & 127;
    const float fac = 1.0f / 300.0f;
    // Barriers are lds_barrier() (s_waitcnt lgkmcnt(0) + s_barrier): __syncthreads() would also wait for vmcnt(0), i.e. for
    // the prefetched window, at the first barrier behind its issue.
    // this lane's packed inputs z[a] = x[2m] + i x[2m+1], m = 128 a + b_: one aligned 4-byte load each.  The workgroup walks
    // jper consecutive symbol steps and always has the NEXT step's eight loads in flight while it transforms the
    // current one: the first version's workgroups all sat through a full memory latency before any arithmetic
    // (6 resident workgroups per CU, lifetime = latency + arithmetic: VALU 65 % busy); now only the first step of a
    // workgroup does.
    const CWSLG_GLOBAL unsigned *d32 = as_global(reinterpret_cast<const unsigned *>(w->frame)) + b_;
    float *const plane = w->spectra;                      // (fetched here: behind the loop's barriers -- memory clobbers -- it was a scalar load and a wait per transform)
    // ---- the slot's finalise, fused (see SpectraFin above): FT8 channels whose frame this boundary has not converted yet
    const bool fuse = FMODE != 0;                                                   // (the host picks the instantiation: every channel of a launch is fused or none is)
    SpectraFin F{};
    v2f cv = {0.0f, 0.0f};                                                          // the pair of samples this lane converts at the top of the next transform ...
    unsigned fin_i = ~0u;                                                           // ... and its index in the frame (>= F.s_end: none)
    unsigned fin_lim = 0u;                                                          // min(n_valid, s_end): pairs at and beyond it are zeros, not loaded
    constexpr int LOOK = 3;                           // windows converted ahead of the one being transformed (>= 3: see "Visibility" above; 6 / 12 / 24 measured: no difference)
    unsigned ring_rd = 0u, ring_wr = 0u;                                            // FMODE 2: this lane's read base / write position in the ring (pairs)
    if (fuse) {
        const FinWork &f = w->fin;
        F.frame = as_global(f.frame); F.out = as_global_rw(f.out); F.nv = f.n_valid;
        const unsigned flen = f.frame_len;
        const float peak = __uint_as_float(*as_global(f.peak));
        float factor = 32767.0f / (peak + 1.0f);
        factor = factor * f.scale;
        F.factor = factor;
        if (blockIdx.x == 0 && tid_ == 0) {
            if (f.peak_next) *as_global_rw(f.peak_next) = 0u;
            if (f.factor_out) *as_global_rw(f.factor_out) = factor;
        }
        // the frame beyond the last window (FT8: samples 180000 .. 239999 of the 20 s frame): only as far as this slot or the previous one put
        // non-zero samples there (FinWork::tail_end; the int16 buffer keeps its zeros otherwise) -- an even share per workgroup of the channel
        const unsigned cover = min((unsigned)(STEP * (nsteps - 1) + NIN), flen), tail_end = min(max(f.tail_end, cover), flen);
        if (tail_end > cover) {
            const unsigned per = ((tail_end - cover + gridDim.x - 1) / gridDim.x + 7u) & ~7u;
            spectra_finalize_range(F, min(cover + per * blockIdx.x, tail_end), min(cover + per * (blockIdx.x + 1), tail_end));
        }
        if constexpr (FMODE == 1) {
            F.s_end = min(((unsigned)(STEP * (jend - 1) + NIN) + 63u) & ~63u, flen);
            // The first LOOK windows now; then 512 samples (one pair per lane) at the top of every transform: after transform j's share the frame is final up
            // to e0 + 512 (j - j0 + 1) >= the line-rounded end of window j + LOOK (a window advances by STEP = 480 <= 512 samples).  n_valid is even (pushes
            // are whole multiples of four blocks), so a pair never straddles it.
            const unsigned e0 = min(((unsigned)(STEP * (j0 + LOOK - 1) + NIN) + 63u) & ~63u, F.s_end);
            spectra_finalize_range(F, (unsigned)(STEP * j0) & ~63u, e0);
            __builtin_amdgcn_s_waitcnt(0x0F70);               // vmcnt(0): this wave's stores have reached the L2 ...
            __syncthreads();                                  // ... and so have every other wave's, before any window is read back
            fin_lim = min(F.nv & ~1u, F.s_end);
            fin_i = e0 + 2u * (unsigned)tid_;
        } else {
            // FMODE 2: window j0 goes into the ring (and the frame) now, 960 pairs by 256 lanes; then lanes 0..239 add the 240 pairs of the next window's last
            // step at the top of every transform
            F.s_end = min((unsigned)(STEP * (jend - 1) + NIN), flen);
            fin_lim = min(F.nv & ~1u, F.s_end);
            const unsigned p0 = (unsigned)(STEP / 2) * (unsigned)j0;               // first pair of window j0
            for (unsigned k = (unsigned)tid_; k < (unsigned)NPACK; k += 256u) {
                const unsigned i = 2u * (p0 + k);
                v2f p = {0.0f, 0.0f};
                if (i < fin_lim) p = *reinterpret_cast<const CWSLG_GLOBAL v2f *>(F.frame + i);
                const v2f bi = p * F.factor + 0.5f;
                const unsigned word = __builtin_amdgcn_perm((unsigned)(int)bi.y, (unsigned)(int)bi.x, 0x05040100u);
                if (i < F.s_end) *reinterpret_cast<CWSLG_GLOBAL unsigned *>(F.out + i) = word;
                const unsigned pos = (p0 + k) % (unsigned)RING;
                s_ring[pos] = word; s_ring[pos + RING] = word;
            }
            __syncthreads();
            fin_i = tid_ < STEP / 2 ? (unsigned)(STEP * j0 + NIN) + 2u * (unsigned)tid_ : 0x40000000u;      // the pair transform j0 converts (lanes 240..255: none, ever)
            ring_rd = (p0 + (unsigned)b_) % (unsigned)RING;
            ring_wr = (p0 + (unsigned)NPACK + (unsigned)tid_) % (unsigned)RING;
        }
        if (fin_i < fin_lim) cv = *reinterpret_cast<const CWSLG_GLOBAL v2f *>(F.frame + fin_i);
    }
    unsigned raw[AMAX];
    if constexpr (FMODE != 2) {
#pragma unroll
        for (int a = 0; a < AMAX; ++a) raw[a] = (128 * a + b_ < NPACK) ? d32[(STEP / 2) * j0 + 128 * a] : 0u;
    } else {                         // window j0 from the ring; from here on ring_rd is the base of the NEXT window, read behind each transform's first barrier
#pragma unroll
        for (int a = 0; a < AMAX; ++a) raw[a] = s_ring[ring_rd + 128 * a];
        ring_rd += (unsigned)(STEP / 2); if (ring_rd >= (unsigned)RING) ring_rd -= (unsigned)RING;
    }
    float2 wn[WINDOW ? AMAX : 1];
    if (WINDOW) {
#pragma unroll
        for (int a = 0; a < AMAX; ++a) {
            const int m = 128 * a + b_;
            wn[WINDOW ? a : 0] = *reinterpret_cast<const float2 *>(tb.win + 2 * ((m < NPACK) ? m : 0));
        }
    }
    Stage1Tw<NA> tw1;
    if (tid_ < 128) stage1_load_tw<0, NA>(tb.w1920, b_, tw1);
    else stage1_load_tw<1, NA>(tb.w1920, b_, tw1);
    // "sparse upper half" (FT8's default search: the stored row ends at most 32 bins above NA * 64 = 960).  The last stage works on ITEMS: a
    // thread takes the butterfly pair P1 = (row r, column q), P2 = (row NA - r, column 63 - q) and unpacks the bins K1 = NA q + r and
    // K2 = NA (63 - q) + NA - r from their results.  512 item slots (8 rows x 64 columns, two per thread), of which row 0 -- which pairs with
    // itself -- fills only columns 0..32.  Round 4: the 31 idle lanes of row 0 (wave 0, columns 33..63) take the 31 bins 961..991 ABOVE 960
    // (kind 2: K = NA (q + 64) + r pairs v1 with u2; kind 3: K = NA (127 - q) + NA - r pairs v2 with u1), and lane 0 takes the two self-paired
    // bins 0 and 960 (kind 1) -- by operand SELECTS inside the same instruction stream, so that every wave runs exactly two items.  Round 3
    // gave the upper bins to one wave as a section of its own behind the item loop and the self-paired bins to a divergent branch of wave 0:
    // s_memtime stamps (scripts/gpu_stamps_spectra.py) showed those two waves 230 and 320 cycles behind the other two at the barrier of
    // every transform.  Same butterflies, same unpack on the same operands: the same bits.
    const bool sparse = nbins >= NA * 64 && nbins <= NA * 64 + 32 && NH == 7;          // wave-uniform (kernel argument)
    constexpr unsigned PW_DUMMY = 4u * (NIN + 31);         // byte offset of a float of s_pw that no stored row reaches in sparse mode
    // LDS byte offsets of the last stage's operands and of pass B's eight points, computed ONCE (from the real thread index): the
    // opaque copy below keeps everything else out of registers, but these are worth theirs -- recomputed per transform they were
    // ~90 of a wave's ~610 VALU instructions (swizzle, row pitch products, shifts).
    float2 w3[IPT][2];
    // CWSLG_SPEC_UNHOIST=1 (measured alternative, round 6): the addresses are recomputed in every transform from the opaque thread index instead of being held
    // in 20 registers across the loop -- the route to a fifth wave per SIMD that needs no hand-written transform (with -DCWSLG_SPEC_WAVES=5)
#ifndef CWSLG_SPEC_UNHOIST
#define CWSLG_SPEC_UNHOIST 0
#endif
    struct ItemAddr { unsigned au1[IPT], au2[IPT], aw1[IPT], aw2[IPT], ap1[IPT], ap2[IPT], aB[8]; bool iskip[IPT], izero[IPT]; };
    auto item_addr = [&](int tidx, ItemAddr &A, int (&K1o)[IPT], int (&K2o)[IPT]) {
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        const int it = tidx + 256 * i;
        int r = it >> 6, q = it & 63, kind = 0;
        bool skip = it >= NITEM || (r == 0 && q > 32);
        int Kup = 0;
        if (sparse && i == 0 && r == 0 && q > 32) {        // wave 0, lanes 33..63: the bin Kup = 928 + q above NA * 64
            Kup = NA * 64 - 32 + q;
            const int d = Kup - NA * 64, a = d / NA, rem = d % NA;
            if (rem <= NH) { kind = 2; r = rem; q = a; }
            else { kind = 3; r = NA - rem; q = 63 - a; }
            skip = Kup >= nbins;
        }
        const int r2 = (r == 0) ? 0 : NA - r;
        const int k2 = (r == 0) ? ((64 - q) & 63) : 63 - q;
        A.izero[i] = r == 0 && q == 0 && kind == 0;
        if (sparse && A.izero[i]) kind = 1;
        A.iskip[i] = skip;
        const int rr = skip ? 0 : r, rr2 = skip ? 0 : r2;
        // the partner column k + 64 is not kept: sy_col(k + 64) = 64 + (sy_col(k) ^ 8) for k < 64, and a row starts at a multiple of 128 bytes,
        // so its byte address is (address of column k ^ 64) + 512 -- one XOR at the point of use, the 512 rides in the instruction's offset field
        static_assert((SY_PITCH * 8) % 128 == 0, "row pitch");
        A.au1[i] = 8u * (unsigned)(rr * SY_PITCH + sy_col(q));
        A.au2[i] = 8u * (unsigned)(rr2 * SY_PITCH + sy_col(k2));
        A.aw1[i] = 8u * (unsigned)q; A.aw2[i] = 8u * (unsigned)k2;
        int K1 = NA * q + r, K2 = NA * k2 + r2;            // the bins of a plain item
        unsigned p1 = 4u * (unsigned)K1, p2 = 4u * (unsigned)K2;
        if (kind == 1) { K2 = NA * 64; p2 = (K2 < nbins) ? 4u * (unsigned)K2 : PW_DUMMY; }        // bins 0 and NA * 64, each paired with itself
        if (kind >= 2) { K1 = Kup; p1 = 4u * (unsigned)Kup; K2 = nbins; p2 = PW_DUMMY; }          // one bin; the second unpack lands in the dummy
        A.ap1[i] = p1 | ((unsigned)kind << 16); A.ap2[i] = p2;
        K1o[i] = (!skip && K1 < nbins) ? K1 : -1; K2o[i] = (!skip && K2 < nbins) ? K2 : -1;
    }
    {
        const int c = tidx >> 4, g = tidx & 15, blk = g >> 3, r = g & 7;
#pragma unroll
        for (int q = 0; q < 8; ++q) A.aB[q] = 8u * (unsigned)((tidx < NGRP ? c : 0) * SY_PITCH + sy_col(64 * blk + r + 8 * q));
    }
    };
    ItemAddr IA0;
    {
        int K1o[IPT], K2o[IPT];
        item_addr(tid_, IA0, K1o, K2o);
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            w3[i][0] = K1o[i] >= 0 ? tb.w3840[K1o[i]] : make_float2(0.f, 0.f);
            w3[i][1] = K2o[i] >= 0 ? tb.w3840[K2o[i]] : make_float2(0.f, 0.f);
        }
    }
    const bool wave0_ = __builtin_amdgcn_readfirstlane(tid_ >> 6) == 0;
    // (the twiddles of the two self-paired bins 0 and NA * 64 are per-lane registers like every other item's -- w3[0][] of lane 0: fetched inside
    // the loop, as in round 3, they were a global load on wave 0's path in EVERY transform whose vmcnt wait also drained the prefetched window:
    // ~550 cycles that the other three waves then spent at the barrier; s_memtime stamps, scripts/gpu_stamps_spectra.py)
    if (tid_ >= 64 && tid_ < 128) s_w128[tid_ - 64] = tb.w128[tid_ - 64];
    // every loop-invariant load (twiddles, window) is waited for HERE, with the builtin the compiler's wait-count pass
    // understands: otherwise it keeps conservative vmcnt waits for them inside the loop (the first iteration could still
    // need them), and from the second iteration on those waits drain the prefetch in the middle of stage 1
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0), expcnt and lgkmcnt untouched

    for (int j = j0; j < jend; ++j) {
    // an opaque copy of the thread index: otherwise every LDS address of the body is hoisted out of the loop and held in
    // registers across it (168 VGPRs, 3 workgroups per CU)
    int t = tid_;
    asm volatile("" : "+v"(t));
    const int tid = t, b = t & 127;
    ItemAddr IAj;
    int k1j[IPT], k2j[IPT];
    if (CWSLG_SPEC_UNHOIST) item_addr(tid, IAj, k1j, k2j);
    const ItemAddr &IA = CWSLG_SPEC_UNHOIST ? IAj : IA0;
    char *const sy_bytes = reinterpret_cast<char *>(&s_y[0][0]);
    const char *const w128_bytes = reinterpret_cast<const char *>(&s_w128[0]);
    char *const pw_bytes = reinterpret_cast<char *>(&s_pw[0]);
    PSTAMP(0);
    float2 z[AMAX];
#pragma unroll
    for (int a = 0; a < AMAX; ++a) {
        const bool live = 128 * a + b < NPACK;
        float lo = (float)(short)(raw[a] & 0xFFFFu), hi = (float)(short)(raw[a] >> 16);
        if (NA != 15) { lo = fac * lo; hi = fac * hi; }                // FT8 (spec v3): the scale rides in stage 1's twiddle
        if (WINDOW) { lo = lo * wn[WINDOW ? a : 0].x; hi = hi * wn[WINDOW ? a : 0].y; }
        z[a] = live ? make_float2(lo, hi) : make_float2(0.f, 0.f);
    }
    // Memory operations of the iteration.  Order of rounds 2-5: the previous step's power row (16 B per lane) leaves first -- its stores have a whole
    // transform to retire before the top of the next iteration waits for vmcnt(0) -- then the next window is requested.  Round 6 measured the other
    // order (loads first, so that in principle only they are waited for; CWSLG_SPEC_LOADS_FIRST=1): hipcc still emits vmcnt(0) at the window's first
    // use (the stores sit behind branches it cannot count), the stores are then the YOUNGEST operations it waits for, and the kernel takes 5.07
    // against 4.53 ms (profiles/r6_sync_ab.txt).  The fused finalise's 4-byte store per lane is what costs: with it compiled out (wrong frames, timing
    // only) the kernel runs at 4.00 ms, the float loads alone cost nothing; non-temporal stores change nothing.
#ifndef CWSLG_SPEC_LOADS_FIRST
#define CWSLG_SPEC_LOADS_FIRST 0
#endif
#ifndef CWSLG_FUSE_STORE_NT
#define CWSLG_FUSE_STORE_NT 0
#endif
    auto store_prev_row = [&]() {
#ifdef CWSLG_SPEC_DIAG_NOSTORE                             // timing diagnostic only (no spectra leave the kernel): what do the plane's stores cost?
        if (j > j0 && nbins < 0) {
#else
        if (j > j0) {
#endif
            CWSLG_GLOBAL v4f *out4 = reinterpret_cast<CWSLG_GLOBAL v4f *>(as_global_rw(CWSLG_SPEC_HOISTPTR ? plane : w->spectra) + (size_t)(j - 1) * nbins);
            // by waves 2-3 only when stage 1 is the prime-factor form: their half of it is 32 instructions shorter than waves 0-1's (stamps: they
            // waited ~430 cycles at the barrier behind stage 1)
            // (round 6: the row's second 120 float4 from waves 0-1 instead of a second pass of waves 2-3 -- 4.50-4.53 against 4.48-4.51 ms: no gain)
            if (NA == 15) { if (tid >= 128) for (int k4 = tid - 128; 4 * k4 < nbins; k4 += 128) plane_store(out4 + k4, *reinterpret_cast<const v4f *>(s_pw + 4 * k4)); }
            else for (int k4 = tid; 4 * k4 < nbins; k4 += 256) plane_store(out4 + k4, *reinterpret_cast<const v4f *>(s_pw + 4 * k4));
        }
    };
    if (!CWSLG_SPEC_LOADS_FIRST) store_prev_row();
    if (fuse) {       // this transform's share of the finalise: the pair fetched during the previous transform leaves as two int16 samples, the next one is fetched.
        // A wave of this kernel advances at the pace of its dependent chains, ~11 cycles per instruction of ANY kind (round 4's stamps), so what this block
        // costs is partly its instruction count: the first in-loop form -- three evaluations of the range ends, scalar branches, two-sample scalar arithmetic,
        // 46 instructions per wave and transform -- cost as much as converting everything in a prologue; this one keeps the index in a register, advances it by
        // a constant and does the arithmetic on the pair.  What remains (+0.3-0.45 ms per 4096 slots) is that the words stored here are READ BACK by the window
        // loads two transforms later: the same store aimed at memory the kernel never reads costs nothing (4.06 against 4.50 ms), reading the lines by a
        // scalar load before they are written does not help, non-temporal stores do not, windows from an LDS ring cost more; and windows read from OTHER memory
        // this workgroup wrote one to four transforms earlier are just as slow (4.9 ms): reading back what the CU has just written is what is slow (profiles/r6_sync_ab.txt).
        const v2f sc = cv * F.factor;                      // buf[k] *= factor            (v_pk_mul_f32: each half rounded on its own -- the same bits)
        const v2f bi = sc + 0.5f;                          // + 0.5f
        const int q0 = (int)bi.x, q1 = (int)bi.y;          // C truncation toward zero, then narrowed to int16
        const unsigned word = __builtin_amdgcn_perm((unsigned)q1, (unsigned)q0, 0x05040100u);      // (q0 & 0xFFFF) | (q1 << 16)
        if (fin_i < F.s_end) {
            if (CWSLG_FUSE_STORE_NT) __builtin_nontemporal_store(word, reinterpret_cast<CWSLG_GLOBAL unsigned *>(F.out + fin_i));
            else *reinterpret_cast<CWSLG_GLOBAL unsigned *>(F.out + fin_i) = word;
            if constexpr (FMODE == 2) { s_ring[ring_wr] = word; s_ring[ring_wr + RING] = word; }       // the last step of window j + 1 (read at the top of j + 1, behind this transform's barriers)
        }
        if constexpr (FMODE == 2) {
            fin_i += (unsigned)STEP;
            ring_wr += (unsigned)(STEP / 2); if (ring_wr >= (unsigned)RING) ring_wr -= (unsigned)RING;
        } else {
            fin_i += 512u;
        }
        cv = v2f{0.0f, 0.0f};
        if (fin_i < fin_lim) cv = *reinterpret_cast<const CWSLG_GLOBAL v2f *>(F.frame + fin_i);
    }
    {   // the next step's window, in flight during this transform.  Unconditional (the workgroup's last step fetches its own window again: eight loads per
        // jper transforms): under `if (j + 1 < jend)` the registers were a merge of two paths, which hipcc kept as two sets and eight 64-bit moves per transform
        const int jn = CWSLG_SPEC_TIGHT ? min(j + 1, jend - 1) : j + 1;
        if constexpr (FMODE != 2) {
        if (CWSLG_SPEC_TIGHT || j + 1 < jend)
#pragma unroll
        for (int a = 0; a < AMAX; ++a) raw[a] = (128 * a + b < NPACK) ? d32[(STEP / 2) * jn + 128 * a] : 0u;
        }
    }
    if (CWSLG_SPEC_LOADS_FIRST) store_prev_row();

    // stage 1 (wave-uniform split of the outputs between waves 0-1 and waves 2-3)
    if (tid < 128) spectra_stage1_regs<0, NA, AMAX>(z, s_y, tw1, b);
    else spectra_stage1_regs<1, NA, AMAX>(z, s_y, tw1, b);
    PSTAMP(1);
    lds_barrier();
    PSTAMP(2);
    if constexpr (FMODE == 2) {      // the next window from the ring (its last quarter was written at the top of this transform, ahead of the barrier): in registers by the next top
#pragma unroll
        for (int a = 0; a < AMAX; ++a) raw[a] = s_ring[ring_rd + 128 * a];
        ring_rd += (unsigned)(STEP / 2); if (ring_rd >= (unsigned)RING) ring_rd -= (unsigned)RING;
    }

    // stage 2, pass A: DIT stages len = 2,4,8 (see the first version)
#ifndef CWSLG_SPEC_MERGE_AB
#define CWSLG_SPEC_MERGE_AB 1          // round 6: passes A and B under ONE exec-mask region instead of three (115 instead of 128 VGPRs, -0.02 ms; 0 = the A/B partner)
#endif
    const bool inAB = tid < NGRP;
    if (CWSLG_SPEC_MERGE_AB ? inAB : true) {
    {
        const int c = tid >> 4, g = tid & 15;
        const int gb = (int)(__brev((unsigned)g) >> 28);
        float2 e[8];
        if (CWSLG_SPEC_MERGE_AB || inAB) {
            // (one address, the eight columns 16 k3 + gb as instruction offsets: indexed as s_y[c][16 * k3 + gb] hipcc built every address from scratch)
            const char *const pa = sy_bytes + 8u * (unsigned)(c * SY_PITCH + gb);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int k3 = ((k & 1) << 2) | (k & 2) | ((k >> 2) & 1);
                e[k] = CWSLG_SPEC_TIGHT ? *reinterpret_cast<const float2 *>(pa + 128 * k3) : s_y[c][16 * k3 + gb];
            }
        }
        // No workgroup barrier here, nor between pass A and pass B: row c is gathered, rewritten and read again by the SAME sixteen
        // lanes (tid >> 4 == c), i.e. inside one wave, whose LDS operations execute in program order (round 3: two barriers fewer per
        // transform, ~0.1 ms each per 4096 slots).  The fences only keep the compiler from moving LDS accesses across these points.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (CWSLG_SPEC_MERGE_AB || inAB) {
            bfly_one(e[0], e[1]); bfly_one(e[2], e[3]); bfly_one(e[4], e[5]); bfly_one(e[6], e[7]);
            bfly_one(e[0], e[2]); bfly_mj(e[1], e[3]); bfly_one(e[4], e[6]); bfly_mj(e[5], e[7]);
            bfly_one(e[0], e[4]); bfly(e[1], e[5], s_w128[16]); bfly_mj(e[2], e[6]); bfly(e[3], e[7], s_w128[48]);
            // sy_col(8 g + k) = sy_col(8 g) ^ k for k < 8 (the swizzle's low three bits are g's, the row starts at a multiple of 128 bytes): one address, seven XORs
            const unsigned a0 = 8u * (unsigned)(c * SY_PITCH + sy_col(8 * g));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (CWSLG_SPEC_TIGHT) *reinterpret_cast<float2 *>(sy_bytes + (a0 ^ (8u * k))) = e[k];
                else s_y[c][sy_col(8 * g + k)] = e[k];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                 // (see above: pass B reads what lanes of this wave wrote)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PSTAMP(3);
    // pass B: stages len = 16,32,64
    if (CWSLG_SPEC_MERGE_AB || inAB) {
        const int r = tid & 7;
        float2 e[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) e[q] = *reinterpret_cast<const float2 *>(sy_bytes + IA.aB[q]);
        {
            const float2 w0 = s_w128[r * 8];
            bfly(e[0], e[1], w0); bfly(e[2], e[3], w0); bfly(e[4], e[5], w0); bfly(e[6], e[7], w0);
        }
        {
            const float2 w0 = s_w128[r * 4], w1 = s_w128[(r + 8) * 4];
            bfly(e[0], e[2], w0); bfly(e[1], e[3], w1); bfly(e[4], e[6], w0); bfly(e[5], e[7], w1);
        }
        {
            bfly(e[0], e[4], s_w128[r * 2]); bfly(e[1], e[5], s_w128[(r + 8) * 2]);
            bfly(e[2], e[6], s_w128[(r + 16) * 2]); bfly(e[3], e[7], s_w128[(r + 24) * 2]);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float2 *>(sy_bytes + IA.aB[q]) = e[q];
    }
    }
    PSTAMP(4);
    lds_barrier();
    PSTAMP(5);

    // last stage (len = 128) fused with the unpack: see the header of this kernel
    float2 w3j[IPT][2];
    if (CWSLG_SPEC_UNHOIST) {        // the unpack twiddles fetched per transform (L1 / L2 hits; the window prefetch issued at the top of the transform has long landed)
#pragma unroll
        for (int i = 0; i < IPT; ++i) {
            w3j[i][0] = k1j[i] >= 0 ? tb.w3840[k1j[i]] : make_float2(0.f, 0.f);
            w3j[i][1] = k2j[i] >= 0 ? tb.w3840[k2j[i]] : make_float2(0.f, 0.f);
        }
    }
    const float2 (&w3u)[IPT][2] = CWSLG_SPEC_UNHOIST ? w3j : w3;
#pragma unroll
    for (int i = 0; i < IPT; ++i) {
        if (IA.iskip[i]) continue;
        float2 u1 = *reinterpret_cast<const float2 *>(sy_bytes + IA.au1[i]), v1 = *reinterpret_cast<const float2 *>(sy_bytes + (IA.au1[i] ^ 64u) + 512);
        float2 u2 = *reinterpret_cast<const float2 *>(sy_bytes + IA.au2[i]), v2 = *reinterpret_cast<const float2 *>(sy_bytes + (IA.au2[i] ^ 64u) + 512);
        bfly(u1, v1, *reinterpret_cast<const float2 *>(w128_bytes + IA.aw1[i]));
        bfly(u2, v2, *reinterpret_cast<const float2 *>(w128_bytes + IA.aw2[i]));
        if (sparse) {
            float2 A1 = u1, B1 = v2, A2 = u2;
            unsigned p1 = IA.ap1[i];
            if (i == 0 && wave0_) {                        // wave-uniform: row 0's lanes -- plain items, the self-paired pair, the bins above NA * 64
                const unsigned kind = p1 >> 16;
                p1 &= 0xFFFFu;
                A1 = (kind == 2) ? v1 : (kind == 3) ? v2 : u1;
                B1 = (kind == 0) ? v2 : (kind == 2) ? u2 : u1;
                A2 = (kind == 1) ? v1 : u2;
            }
            *reinterpret_cast<float *>(pw_bytes + p1) = unpack_power(A1, B1, w3u[i][0]);
            *reinterpret_cast<float *>(pw_bytes + IA.ap2[i]) = unpack_power(A2, v1, w3u[i][1]);
            continue;
        }
        if (IA.izero[i]) {                                     // Z[0] and Z[NZ/2] pair with themselves
            s_pw[0] = unpack_power(u1, u1, w3u[i][0]);
            if (NA * 64 < nbins) s_pw[NA * 64] = unpack_power(v1, v1, tb.w3840[NA * 64]);
            if (NZ < nbins) s_pw[NZ] = unpack_power(u1, u1, tb.w3840[NZ]);
            continue;
        }
        const int K1 = (int)(IA.ap1[i] >> 2), K2 = (int)(IA.ap2[i] >> 2);
        const int K3 = K1 + NA * 64, K4 = K2 + NA * 64;
        if (K1 < nbins) s_pw[K1] = unpack_power(u1, v2, w3u[i][0]);
        if (K2 < nbins) s_pw[K2] = unpack_power(u2, v1, w3u[i][1]);
        if (K3 < nbins) s_pw[K3] = unpack_power(v1, u2, tb.w3840[K3]);
        if (K4 < nbins) s_pw[K4] = unpack_power(v2, u1, tb.w3840[K4]);
    }
    for (int k = NZ + 1 + tid; k < nbins; k += 256) s_pw[k] = 0.0f;          // padding beyond the Nyquist bin
    PSTAMP(6);
    lds_barrier();
    PSTAMP(7);
    }   // next symbol step: s_y is rewritten after this barrier, s_pw only after four more
    if (jend > j0) {
        CWSLG_GLOBAL v4f *out4 = reinterpret_cast<CWSLG_GLOBAL v4f *>(as_global_rw(w->spectra) + (size_t)(jend - 1) * nbins);
        for (int k4 = tid_; 4 * k4 < nbins; k4 += 256) plane_store(out4 + k4, *reinterpret_cast<const v4f *>(s_pw + 4 * k4));
    }
}

constexpr int SYNC2D_NT = 512;          // 8 waves: one bin per wave at a time, 4 bins per wave per band
// ---------------------------------------------------------------------------------------------
// Costas correlation + lag peak search.  grid (ceil((ib-ia+1)/32), n_channels), 256 threads.
// LDS: the band's 44 spectrum rows [44][376] (66 KB) + per-bin 7-tone sums for the 2 bins in flight => 2 WG/CU.
// Per bin: a wave pair, lane = lag (125 lags), 42 conflict-free LDS reads per lane; the +-10 and +-62 peak
// searches are wavefront arg-max reductions on an order-preserving 64-bit key (value, -lag).
__device__ __forceinline__ unsigned long long sync_key(float v, int l)
{
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);              // total order of floats as unsigned
    return ((unsigned long long)u << 32) | (unsigned)(255 - l);  // ties: the smaller lag wins (maxloc: first maximum)
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long k)
{
    const int lo = __builtin_amdgcn_update_dpp((int)(unsigned)k, (int)(unsigned)k, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(unsigned)(k >> 32), (int)(unsigned)(k >> 32), CTRL, 0xf, 0xf, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
// wavefront max of a 64-bit key; every lane ends with the maximum.  DPP inside the 16-lane rows
// (quad xor 1, quad xor 2, half-row mirror, row mirror), ds_bpermute across rows.
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long k)
{
    unsigned long long o;
    o = dpp_u64<0xB1>(k); k = (o > k) ? o : k;      // quad_perm [1,0,3,2]
    o = dpp_u64<0x4E>(k); k = (o > k) ? o : k;      // quad_perm [2,3,0,1]
    o = dpp_u64<0x141>(k); k = (o > k) ? o : k;     // row_half_mirror
    o = dpp_u64<0x140>(k); k = (o > k) ? o : k;     // row_mirror
#pragma unroll
    for (int msk = 16; msk <= 32; msk <<= 1) {
        const unsigned lo = __shfl_xor((unsigned)k, msk, 64);
        const unsigned hi = __shfl_xor((unsigned)(k >> 32), msk, 64);
        o = ((unsigned long long)hi << 32) | lo;
        k = (o > k) ? o : k;
    }
    return k;
}
__device__ __forceinline__ float key_value(unsigned long long k)
{
    unsigned u = (unsigned)(k >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    return __uint_as_float(u);
}

// sync2d(i, j) and sync2d(i, j + 1) for one bin (band row rr): two adjacent lags per lane.  The LDS images are indexed by
// the 1-based symbol step m itself (column 0 unused), so the pair's operands (m, m + 1) with m = j + 12 + 4n (+144, +288)
// even-aligned are ONE 8-byte read: half the LDS instructions of a lag-per-read search and ds_read_b64's 256 B/clk instead
// of ds_read_b32's 128.  A term the restatement skips (m < 1 or m > NHSYM) is read from a clamped address and replaced by
// +0.0: adding +0.0 to a non-negative float sum is exact, so the sums are the restatement's bit for bit.
#if CWSLG_LAB      // round 1's Costas search (CWSLG_SYNC_VARIANT bit 0), lab library only
struct SyncPair { float a, b; };
__device__ __forceinline__ SyncPair costas_sync2(const float (*s_s)[376], const float *s_c0, int rr, int j)
{
    const int icos[7] = {3, 1, 4, 0, 6, 5, 2};
    float ta0 = 0, tb0 = 0, tc0 = 0, ua0 = 0, ub0 = 0, uc0 = 0;      // lag j:     t sums, t0 sums
    float ta1 = 0, tb1 = 0, tc1 = 0, ua1 = 0, ub1 = 0, uc1 = 0;      // lag j + 1
#pragma unroll
    for (int n = 0; n < 7; ++n) {
        const int m = j + 12 + 4 * n;                  // even; lag j + 1 uses m + 1
        const float *row = s_s[rr + 2 * icos[n]];
        {   // block a: valid iff 1 <= m <= NHSYM
            const int mc = (m < 0) ? 0 : m;
            const float2 v = *reinterpret_cast<const float2 *>(row + mc), c = *reinterpret_cast<const float2 *>(s_c0 + mc);
            const bool ok0 = m >= 1, ok1 = m + 1 >= 1;                // m + 1 <= 62 + 12 + 24 + 1 < NHSYM always
            ta0 = ta0 + (ok0 ? v.x : 0.0f); ua0 = ua0 + (ok0 ? c.x : 0.0f);
            ta1 = ta1 + (ok1 ? v.y : 0.0f); ua1 = ua1 + (ok1 ? c.y : 0.0f);
        }
        {   // block b: always inside the frame
            const float2 v = *reinterpret_cast<const float2 *>(row + m + 144), c = *reinterpret_cast<const float2 *>(s_c0 + m + 144);
            tb0 = tb0 + v.x; ub0 = ub0 + c.x;
            tb1 = tb1 + v.y; ub1 = ub1 + c.y;
        }
        {   // block c: valid iff m + 288 <= NHSYM
            const int mm = m + 288, mc = (mm > 374) ? 374 : mm;
            const float2 v = *reinterpret_cast<const float2 *>(row + mc), c = *reinterpret_cast<const float2 *>(s_c0 + mc);
            const bool ok0 = mm <= FT8_NHSYM, ok1 = mm + 1 <= FT8_NHSYM;
            tc0 = tc0 + (ok0 ? v.x : 0.0f); uc0 = uc0 + (ok0 ? c.x : 0.0f);
            tc1 = tc1 + (ok1 ? v.y : 0.0f); uc1 = uc1 + (ok1 ? c.y : 0.0f);
        }
    }
    auto finish = [](float ta, float tb, float tc, float t0a, float t0b, float t0c) {
        float t = ta + tb + tc;
        float t0 = t0a + t0b + t0c;
        t0 = (t0 - t) / 6.0f;
        const float sync_abc = t / t0;
        t = tb + tc;
        t0 = t0b + t0c;
        t0 = (t0 - t) / 6.0f;
        const float sync_bc = t / t0;
        float sy = (sync_abc > sync_bc) ? sync_abc : sync_bc;
        if (!(sy == sy)) sy = 0.0f;                    // 0/0 on all-zero windows: defined as 0 (as the oracle)
        return sy;
    };
    return SyncPair{finish(ta0, tb0, tc0, ua0, ub0, uc0), finish(ta1, tb1, tc1, ua1, ub1, uc1)};
}

__global__ __launch_bounds__(SYNC2D_NT) void ft8_sync2d_kernel(const SyncWork *__restrict__ works, int ia, int ib, int nbins)
{
    constexpr int ROWS = SYNC_BAND + 12, PITCH = 376;
    __shared__ __attribute__((aligned(16))) float s_s[ROWS][PITCH];              // s_s[r][m] = s(i0 + r, m), m = 1..NHSYM
    __shared__ __attribute__((aligned(16))) float s_c0[SYNC2D_NT / 64][PITCH];
    const SyncWork *w = works + blockIdx.y;
    const int i0 = ia + blockIdx.x * SYNC_BAND;
    const int tid = threadIdx.x;
    // stage the band; loads batched 16 deep so their latencies overlap
    {
        const float *sp = w->spectra;
        constexpr int TOTAL = ROWS * FT8_NHSYM;            // 16368
        for (int e0 = 0; e0 < TOTAL; e0 += SYNC2D_NT * 16) {
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + q * SYNC2D_NT + tid;
                const int m = e / ROWS, r = e - m * ROWS;
                const int bin = i0 + r;
                v[q] = (e < TOTAL && bin < nbins) ? sp[(size_t)m * nbins + bin] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int e = e0 + q * SYNC2D_NT + tid;
                const int m = e / ROWS, r = e - m * ROWS;
                if (e < TOTAL) s_s[r][m + 1] = v[q];
            }
        }
        if (tid < ROWS) { s_s[tid][0] = 0.0f; s_s[tid][373] = 0.0f; s_s[tid][374] = 0.0f; s_s[tid][375] = 0.0f; }
    }
    __syncthreads();

    // one wave per bin from here on: no workgroup barriers, each wave owns s_c0[wv]
    const int lane = tid & 63, wv = tid >> 6;
    float *c0 = s_c0[wv];
    for (int rr = wv; rr < SYNC_BAND; rr += SYNC2D_NT / 64) {
        const int bin = i0 + rr;
        if (bin > ib) break;                            // wave-uniform
        // 7-tone sums of this bin for every symbol step (sequential k, as the restatement), two steps per lane and read
        for (int mp = lane; mp < 188; mp += 64) {       // columns (2 mp, 2 mp + 1): 0..375
            float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const float2 v = *reinterpret_cast<const float2 *>(&s_s[rr + 2 * k][2 * mp]);
                acc.x = acc.x + v.x;
                acc.y = acc.y + v.y;
            }
            *reinterpret_cast<float2 *>(c0 + 2 * mp) = acc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // lags: lane -> the pair (j, j + 1), j = 2 lane - 62: -62 .. 62 in lanes 0..62 (the pair's second lag 63 does not exist)
        const int j = 2 * lane - FT8_JZ;
        const bool ok0 = j <= FT8_JZ, ok1 = j + 1 <= FT8_JZ;
        SyncPair sy = SyncPair{0.0f, 0.0f};
        if (ok0) sy = costas_sync2(s_s, c0, rr, j);
        const unsigned long long ka = ok0 ? sync_key(sy.a, j + FT8_JZ) : 0ull;
        const unsigned long long kb = ok1 ? sync_key(sy.b, j + 1 + FT8_JZ) : 0ull;
        unsigned long long k2 = (kb > ka) ? kb : ka;
        unsigned long long k1 = 0ull;
        if (ok0 && j >= -10 && j <= 10) k1 = ka;
        if (ok1 && j + 1 >= -10 && j + 1 <= 10 && kb > k1) k1 = kb;
        k2 = wave_max_u64(k2);
        k1 = wave_max_u64(k1);
        if (lane == 0) {
            w->red[bin] = key_value(k1);  w->jpeak[bin] = (255 - (int)(k1 & 0xFFu)) - FT8_JZ;
            w->red2[bin] = key_value(k2); w->jpeak2[bin] = (255 - (int)(k2 & 0xFFu)) - FT8_JZ;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();               // c0 is rewritten by the next bin
    }
}

#endif  // CWSLG_LAB
// ---------------------------------------------------------------------------------------------
// ft8_sync2d_v2_kernel: the same search, same sums in the same order (bit-identical results), restructured after the
// round-1 counters (LDS 53 % busy with 28 % of it bank conflicts, 2232 VALU instructions per wave):
//   * band image s_s[44][378], column = symbol step m + 2, with REAL zero columns for m = -2, -1, 0 and m = 373..375:
//     a term the restatement skips is read from a clamped address that holds +0.0 (adding +0.0 to a non-negative sum is
//     exact), so the 56 per-lane selects of the first version are gone;
//   * staging: wave w takes steps w, w+8, ...; lanes 0..43 load the band's 44 bins of one step (one 176-byte run) and write
//     them down a column: row pitch 378 = 26 (mod 32) puts the 32 lanes of a write group on 16 banks (2-way: free for
//     ds_write_b32) where pitch 376 put them on 4 (8-way); no per-element division by the row count;
//   * peak searches: float wavefront maximum by DPP (4 steps inside the 16-lane rows, 4 readlanes across them), then a
//     ballot picks the FIRST lane holding it -- `first || sy > best` of the restatement, including its -0 == +0 -- instead
//     of two reductions over 64-bit (value, lag) keys.
constexpr int S2_PITCH = 378, S2_COL0 = 2;
constexpr int DPP_ROW_MIRROR = 0x140;

__device__ __forceinline__ float wave_max_f32(float v)
{
    v = fmaxf(v, dpp_get<DPP_XOR1>(v));
    v = fmaxf(v, dpp_get<DPP_XOR2>(v));
    v = fmaxf(v, dpp_get<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_get<DPP_ROW_MIRROR>(v));                   // every lane of a 16-lane row holds the row maximum
    const int vi = __float_as_int(v);                           // readlane moves bits: the builtin is typed int
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// first lane (lowest lag) whose value equals the wavefront maximum; lanes that do not take part pass -inf
__device__ __forceinline__ void wave_first_max(float v, int lag, float &best, int &best_lag)
{
    const float m = wave_max_f32(v);
    const unsigned long long hit = __builtin_amdgcn_ballot_w64(v == m);
    const int src = hit ? (int)__builtin_ctzll(hit) : 0;
    best = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
    best_lag = __builtin_amdgcn_readlane(lag, src);
}

// The search's two reductions of a bin (+-62 over every lag, +-10 over the near ones) together, as fused DPP maxima (v_max_f32_dpp: permute and
// compare in one instruction) -- rows of 16 by quad_perm / row_half_mirror / row_mirror, then row_bcast:15 / :31 into the last lane.  hipcc's form of
// wave_first_max is a move, a DPP move and two v_max (one to canonicalise) per step plus four v_readlane and three more maxima: ~36 issue slots per
// reduction, 72 of a bin's ~340; here 2 x 13.  The two chains interleave (a DPP read of a register wants two wait states behind its VALU write).  No
// NaN reaches this (sync_finish returns 0 for 0 / 0), a maximum is exact: the same values, the same first lane.
#ifndef CWSLG_SEARCH_HOISTPTR
#define CWSLG_SEARCH_HOISTPTR 1          // the four result pointers fetched once per band (0: inside the bin loop, where the statement's memory clobber makes them a scalar load and a wait per bin)
#endif
#ifndef CWSLG_SEARCH_DPPMAX
#define CWSLG_SEARCH_DPPMAX 1          // 0: hipcc's wave_first_max twice (the A/B partner: scripts/gpu_r5_dppmax.sh)
#endif
__device__ __forceinline__ void wave_first_max2(float va, int laga, float vb, int lagb, float &besta, int &best_laga, float &bestb, int &best_lagb)
{
    float ma = va, mb = vb;
#define CWSLG_DPP_MAX2(CTRL)                                                                  \
    "v_max_f32_dpp %0, %0, %0 " CTRL "\n\t"                                                   \
    "v_max_f32_dpp %1, %1, %1 " CTRL "\n\t"                                                   \
    "s_nop 0\n\t"
    asm volatile("s_nop 1\n\t"
                 CWSLG_DPP_MAX2("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 CWSLG_DPP_MAX2("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 CWSLG_DPP_MAX2("row_half_mirror row_mask:0xf bank_mask:0xf")
                 CWSLG_DPP_MAX2("row_mirror row_mask:0xf bank_mask:0xf")
                 CWSLG_DPP_MAX2("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 CWSLG_DPP_MAX2("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 0"
                 : "+v"(ma), "+v"(mb));
#undef CWSLG_DPP_MAX2
    const float wa = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ma), 63));
    const float wb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mb), 63));
    const unsigned long long hita = __builtin_amdgcn_ballot_w64(va == wa), hitb = __builtin_amdgcn_ballot_w64(vb == wb);
    const int srca = hita ? (int)__builtin_ctzll(hita) : 0, srcb = hitb ? (int)__builtin_ctzll(hitb) : 0;
    // (the value is read from that lane too: a maximum of zero can be held with either sign)
    besta = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(va), srca)); best_laga = __builtin_amdgcn_readlane(laga, srca);
    bestb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vb), srcb)); best_lagb = __builtin_amdgcn_readlane(lagb, srcb);
}

// x / 6.0f, correctly rounded, in three operations instead of the ten of the IEEE division sequence: q0 = x * fl(1/6), the residual
// x - 6 q0 is exact in one fmaf, one more fmaf corrects q0.  Checked against the division for EVERY float (tests/div6_check.c,
// tests/test_div6_shortcut.py): identical for all x with 2^-125 <= |x| < inf and for +-0; below that range the quotient is denormal
// (double rounding) and at inf the residual is NaN -- those inputs (with margin: everything below 2^-95, zero included) take the division itself.
__device__ __forceinline__ float div6_exact(float x)
{
    const float r = 0x1.555556p-3f;                     // fl(1/6)
    const float q0 = x * r;
    const float e = __builtin_fmaf(-6.0f, q0, x);
    float q = __builtin_fmaf(e, r, q0);
    // |x| outside [2^-95, FLT_MAX] -- zero included: the division gives the same +-0 -- takes the division (one compare on the magnitude bits)
    if (__builtin_expect((__float_as_uint(x) & 0x7fffffffu) - 0x10000000u >= 0x6f800000u, 0)) q = x / 6.0f;
    return q;
}

__device__ __forceinline__ float sync_finish(float ta, float tb, float tc, float t0a, float t0b, float t0c)
{
    float t = ta + tb + tc;
    float t0 = t0a + t0b + t0c;
    t0 = div6_exact(t0 - t);
    const float sync_abc = t / t0;
    t = tb + tc;
    t0 = t0b + t0c;
    t0 = div6_exact(t0 - t);
    const float sync_bc = t / t0;
    float sy = (sync_abc > sync_bc) ? sync_abc : sync_bc;
    if (!(sy == sy)) sy = 0.0f;                        // 0/0 on all-zero windows: defined as 0 (as the oracle)
    return sy;
}

// sync_finish for a lane's two lags at once: the four x / 6 share ONE range test (the maximum of the four biased magnitudes against the bound: 11 vector
// instructions and one branch where four div6_exact take 12 and four); a lane outside the range -- never, on audio -- redoes all four by division, which
// gives what the short form gives wherever that is valid.  Same operations on the same operands: the same bits as sync_finish twice.
__device__ __forceinline__ void sync_finish2(v2f ta, v2f tb, v2f tc, v2f ua, v2f ub, v2f uc, float &sx, float &sy)
{
    const float r = 0x1.555556p-3f;                     // fl(1/6)
    const float t1x = ta.x + tb.x + tc.x, t1y = ta.y + tb.y + tc.y;
    const float d1x = (ua.x + ub.x + uc.x) - t1x, d1y = (ua.y + ub.y + uc.y) - t1y;
    const float t2x = tb.x + tc.x, t2y = tb.y + tc.y;
    const float d2x = (ub.x + uc.x) - t2x, d2y = (ub.y + uc.y) - t2y;
    auto short6 = [&](float x) { const float q0 = x * r; return __builtin_fmaf(__builtin_fmaf(-6.0f, q0, x), r, q0); };
    auto biased = [](float x) { return (__float_as_uint(x) & 0x7fffffffu) - 0x10000000u; };
    float q1x = short6(d1x), q1y = short6(d1y), q2x = short6(d2x), q2y = short6(d2y);
    const unsigned worst = max(max(biased(d1x), biased(d1y)), max(biased(d2x), biased(d2y)));
    if (__builtin_expect(worst >= 0x6f800000u, 0)) { q1x = d1x / 6.0f; q1y = d1y / 6.0f; q2x = d2x / 6.0f; q2y = d2y / 6.0f; }
    const float ax = t1x / q1x, bx = t2x / q2x, ay = t1y / q1y, by = t2y / q2y;
    sx = (ax > bx) ? ax : bx;
    sy = (ay > by) ? ay : by;
    if (!(sx == sx)) sx = 0.0f;                        // 0/0 on all-zero windows: defined as 0 (as the oracle)
    if (!(sy == sy)) sy = 0.0f;
}

#if defined(CWSLG_STAMP) && defined(CWSLG_STAMP_SYNC)
// diagnostic build only (scripts/gpu_stamps_sync.py): s_memtime of wave 0 at the phase seams of the workgroup's staging and FIRST bin
#define SSTAMP(slot)                                                                                \
    do {                                                                                            \
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                   \
        if (threadIdx.x == 0 && wg_ < 65536) {                                                      \
            unsigned long long t_;                                                                  \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            g_stamps[8 * wg_ + (slot)] = t_;                                                        \
        }                                                                                           \
    } while (0)
// ... without waiting for outstanding global loads (a prefetch is in flight)
#define SSTAMP1(slot)                                                                               \
    do {                                                                                            \
        const unsigned wg_ = blockIdx.y * gridDim.x + blockIdx.x;                                   \
        if (threadIdx.x == 0 && wg_ < 65536) {                                                      \
            unsigned long long t_;                                                                  \
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            g_stamps[8 * wg_ + (slot)] = t_;                                                        \
        }                                                                                           \
    } while (0)
#else
#define SSTAMP(slot) do { } while (0)
#define SSTAMP1(slot) do { } while (0)
#endif
#if CWSLG_LAB      // round 2's search (CWSLG_SYNC_VARIANT bit 6), lab library only
__global__ __launch_bounds__(SYNC2D_NT, 4) void ft8_sync2d_v2_kernel(const SyncWork *__restrict__ works, int ia, int ib, int nbins)
{
    constexpr int ROWS = SYNC_BAND + 12, NW = SYNC2D_NT / 64;
    constexpr int UN = (FT8_NHSYM + NW - 1) / NW;                              // symbol steps per wave: 47
    __shared__ __attribute__((aligned(16))) float s_s[ROWS][S2_PITCH];       // s_s[r][m + 2] = s(i0 + r, m)
    __shared__ __attribute__((aligned(16))) float s_c0[NW][S2_PITCH];
    const SyncWork *w = works + blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = ia + blockIdx.x * SYNC_BAND;
    {   // every load of the wave is issued before the first LDS write: one memory latency per band (with batches of 8 the
        // workgroup paid six in a row, and only two workgroups fit a CU to cover for each other)
        const CWSLG_GLOBAL float *sp = as_global(w->spectra) + i0 + lane;
        const bool on = lane < ROWS && i0 + lane < nbins;
        float v[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int m = wv + NW * q;
            v[q] = (on && m < FT8_NHSYM) ? sp[(size_t)m * nbins] : 0.0f;
        }
        float *dst = &s_s[lane < ROWS ? lane : 0][S2_COL0 + 1 + wv];           // column of step m = 1 + (0-based step)
#pragma unroll
        for (int q = 0; q < UN; ++q)
            if (lane < ROWS && wv + NW * q < FT8_NHSYM) dst[NW * q] = v[q];
        if (tid < ROWS) {
            float *row = s_s[tid];
            row[0] = 0.0f; row[1] = 0.0f; row[2] = 0.0f;                        // m = -2, -1, 0
            row[S2_COL0 + 373] = 0.0f; row[S2_COL0 + 374] = 0.0f; row[S2_COL0 + 375] = 0.0f;
        }
    }
    __syncthreads();
    // (A persistent form that walks several bands and keeps the next band's 47 loads per lane in registers was tried: the
    // search itself needs ~100 VGPRs at this occupancy, the prefetch spills, and hipcc may not spill an in-flight load.)
    float *c0 = s_c0[wv];
    const int icos[7] = {3, 1, 4, 0, 6, 5, 2};
    const int j = 2 * lane - FT8_JZ;                       // this lane's lag pair (j, j + 1); lane 63 has none
    const bool ok0 = j <= FT8_JZ, ok1 = j + 1 <= FT8_JZ;
    const bool near0 = j >= -10 && j <= 10, near1 = j + 1 >= -10 && j + 1 <= 10;
    const float ninf = -__builtin_huge_valf();
    // Column PAIRS: lag pair (j, j + 1) reads the columns (m + 2, m + 3), m = j + 12 + 4 n (+144, +288) even, i.e. float2 number
    // lane - 24 + 2 n (+72, +144) of the row.  The images are indexed as float2 arrays throughout: indexed as float arrays with an
    // even offset, hipcc cannot prove the 8-byte alignment and splits every access into two ds_read_b32 (round-2 counters: 39 % of
    // the LDS cycles were bank conflicts from exactly that).
    const int pa0 = lane - 24, pb0 = pa0 + 72, pc0 = pa0 + 144;
    constexpr int PP = S2_PITCH / 2;                       // float2 per row
    // (v2f is the built-in vector type: one <2 x float> load = one ds_read_b64; HIP's float2 is a struct whose copy hipcc scalarises)
    const v2f *s_s2 = reinterpret_cast<const v2f *>(&s_s[0][0]);
    v2f *c02 = reinterpret_cast<v2f *>(c0);
    {
        // one wave per bin from here on, each wave owns s_c0[wv]
        for (int rr = wv; rr < SYNC_BAND; rr += NW) {
            const int bin = i0 + rr;
            if (bin > ib) break;                            // wave-uniform
            // 7-tone sums of this bin for every symbol step (sequential k, as the restatement), two steps per lane and read
            for (int mp = lane; mp < PP; mp += 64) {
                v2f acc = {0.0f, 0.0f};
#pragma unroll
                for (int k = 0; k < 7; ++k) acc = acc + s_s2[(rr + 2 * k) * PP + mp];      // two steps per lane: one v_pk_add_f32 each
                c02[mp] = acc;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            v2f ta = {0, 0}, tb = {0, 0}, tc = {0, 0}, ua = {0, 0}, ub = {0, 0}, uc = {0, 0};   // t sums, t0 sums of lags (j, j + 1): packed adds
#pragma unroll
            for (int n = 0; n < 7; ++n) {
                const v2f *row = s_s2 + (rr + 2 * icos[n]) * PP;
                int pa = pa0 + 2 * n; pa = pa < 0 ? 0 : pa;                   // m < -2  -> the zero pair (m = -2, -1)
                int pc = pc0 + 2 * n; pc = pc > PP - 1 ? PP - 1 : pc;         // m > 374 -> the zero pair (m = 374, 375)
                const int pb = pb0 + 2 * n;
                const v2f va = row[pa], wa = c02[pa];
                const v2f vb = row[pb], wb = c02[pb];
                const v2f vc = row[pc], wc = c02[pc];
                ta = ta + va; ua = ua + wa;
                tb = tb + vb; ub = ub + wb;
                tc = tc + vc; uc = uc + wc;
            }
            const float sa = ok0 ? sync_finish(ta.x, tb.x, tc.x, ua.x, ub.x, uc.x) : ninf;
            const float sb = ok1 ? sync_finish(ta.y, tb.y, tc.y, ua.y, ub.y, uc.y) : ninf;
            // +-62: the lane's own first maximum (lag j before j + 1), then the wavefront's
            const bool b2 = sb > sa;
            float r2; int l2;
            wave_first_max(b2 ? sb : sa, b2 ? j + 1 : j, r2, l2);
            const float na = near0 ? sa : ninf, nb = near1 ? sb : ninf;
            const bool b1 = nb > na;
            float r1; int l1;
            wave_first_max(b1 ? nb : na, b1 ? j + 1 : j, r1, l1);
            if (lane == 0) {
                as_global_rw(w->red)[bin] = r1;  as_global_rw(w->jpeak)[bin] = l1;
                as_global_rw(w->red2)[bin] = r2; as_global_rw(w->jpeak2)[bin] = l2;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();               // c0 is rewritten by the next bin
        }
    }
}

#endif  // CWSLG_LAB
// ---------------------------------------------------------------------------------------------
// The hand-scheduled band search (round 3).  Round-3 stamps of ft8_sync2d_v2_kernel (scripts/gpu_stamps_sync.py): a workgroup lives
// ~22 700 ticks, 7300 staging and ~3200 per bin, of which the arithmetic is a fraction: the wave sat through EIGHT LDS round trips per
// bin (three for the 7-tone sums c0, whose loop hipcc does not unroll, five for the 42 reads of the search, which it issues in
// batches each closed by lgkmcnt(0)) at ~400 ticks each under load; asked for all reads up front (arrays + sched_barrier) it spills.
// Here one inline-assembly stream per bin (sync2d_asm.inc, generated by scripts/gen_sync2d_asm.py) issues the 42 reads of this bin's
// Costas sums and the 21 reads of the NEXT bin's c0 in order, twelve to fifteen in flight, every group of three adds waiting only for
// its own operands; temporaries live in fixed registers v94..v127.  The first read of every sum lands in the accumulator itself
// (0 + x = x exactly: powers are never -0).  Same band image, same sums in the same order, same peak rules: bit-identical results.
#include "sync2d_asm.inc"
// The search of one band by one wave (bins i0 + wvu, i0 + wvu + NW, ... inside [ia, ib]): shared by ft8_sync2d_v3_kernel and
// ft8_sync_chan_kernel.  s_base = LDS address of the band image (row pitch S2_PITCH floats), sC = LDS address of this wave's c0 row.
template <int NW>
__device__ __forceinline__ void sync2d_search_band(const SyncWork *w, unsigned s_base, unsigned sC, int wvu, int lane, int i0, int ia, int ib)
{
    constexpr int PP = S2_PITCH / 2, RB = S2_PITCH * 4;
    static_assert(S2_PITCH == 378, "sync2d_asm.inc is generated for this row pitch");
    const int j = 2 * lane - FT8_JZ;                       // this lane's lag pair (j, j + 1); lane 63 has none
    const bool ok0 = j <= FT8_JZ, ok1 = j + 1 <= FT8_JZ;
    const bool near0 = j >= -10 && j <= 10, near1 = j + 1 >= -10 && j + 1 <= 10;
    const float ninf = -__builtin_huge_valf();
    // byte offsets of this lane's operand pairs inside a row (lane - 24 + 2 n clamped below to the zero pair 0, lane + 120 + 2 n
    // clamped above to the zero pair PP - 1, see v2) and of its c0 columns
    unsigned vA[7], vC[3];
#pragma unroll
    for (int n = 0; n < 7; ++n) { const int p = lane - 24 + 2 * n; vA[n] = 8u * (unsigned)(p < 0 ? 0 : p); }
#pragma unroll
    for (int n = 4; n < 7; ++n) { const int p = lane + 120 + 2 * n; vC[n - 4] = 8u * (unsigned)(p > PP - 1 ? PP - 1 : p); }
    const unsigned vU = 8u * (unsigned)lane, vL2 = 8u * (unsigned)(lane + 128 > PP - 1 ? PP - 1 : lane + 128);
#if CWSLG_SEARCH_HOISTPTR
    float *const p_red = w->red, *const p_red2 = w->red2;
    int *const p_jpeak = w->jpeak, *const p_jpeak2 = w->jpeak2;
#endif
    int rr = wvu;
    while (rr < SYNC_BAND && i0 + rr < ia) rr += NW;       // (a band may start below the first searched bin)
    bool live = rr < SYNC_BAND && i0 + rr <= ib;           // wave-uniform
    if (live) {
        const unsigned sN = (unsigned)__builtin_amdgcn_readfirstlane((int)(s_base + (unsigned)rr * RB));
        asm volatile(SYNC2D_ASM_C0_ONLY : : [vU] "v"(vU), [vL2] "v"(vL2), [sC] "s"(sC), [sN] "s"(sN) : SYNC2D_ASM_CLOBBERS);
    }
    SSTAMP(2);
    const int rr0 = rr;
    while (live) {
        const int bin = i0 + rr;
        const int rn = rr + NW;
        const bool more = rn < SYNC_BAND && i0 + rn <= ib;               // wave-uniform
        const unsigned sS = (unsigned)__builtin_amdgcn_readfirstlane((int)(s_base + (unsigned)rr * RB));
        const unsigned sN = sS + NW * RB;
        v2f ta, tb, tc, ua, ub, uc;                                      // t sums, t0 sums of lags (j, j + 1)
        if (more)
            asm volatile(SYNC2D_ASM_SEARCH_NEXT
                         : [ta] "=&v"(ta), [ua] "=&v"(ua), [tb] "=&v"(tb), [ub] "=&v"(ub), [tc] "=&v"(tc), [uc] "=&v"(uc)
                         : [vA0] "v"(vA[0]), [vA1] "v"(vA[1]), [vA2] "v"(vA[2]), [vA3] "v"(vA[3]), [vA4] "v"(vA[4]), [vA5] "v"(vA[5]), [vA6] "v"(vA[6]),
                           [vC4] "v"(vC[0]), [vC5] "v"(vC[1]), [vC6] "v"(vC[2]), [vU] "v"(vU), [vL2] "v"(vL2), [sS] "s"(sS), [sC] "s"(sC), [sN] "s"(sN)
                         : SYNC2D_ASM_CLOBBERS);
        else
            asm volatile(SYNC2D_ASM_SEARCH_LAST
                         : [ta] "=&v"(ta), [ua] "=&v"(ua), [tb] "=&v"(tb), [ub] "=&v"(ub), [tc] "=&v"(tc), [uc] "=&v"(uc)
                         : [vA0] "v"(vA[0]), [vA1] "v"(vA[1]), [vA2] "v"(vA[2]), [vA3] "v"(vA[3]), [vA4] "v"(vA[4]), [vA5] "v"(vA[5]), [vA6] "v"(vA[6]),
                           [vC4] "v"(vC[0]), [vC5] "v"(vC[1]), [vC6] "v"(vC[2]), [vU] "v"(vU), [sS] "s"(sS), [sC] "s"(sC)
                         : SYNC2D_ASM_CLOBBERS);
        if (rr == rr0) SSTAMP(3);
        // (computed on every lane and selected afterwards: 62 or 63 of the 64 lanes hold a lag, a branch around the divisions saves nothing)
        float sa, sb;
#if CWSLG_SEARCH_DPPMAX
        sync_finish2(ta, tb, tc, ua, ub, uc, sa, sb);
#else
        sa = sync_finish(ta.x, tb.x, tc.x, ua.x, ub.x, uc.x);
        sb = sync_finish(ta.y, tb.y, tc.y, ua.y, ub.y, uc.y);
#endif
        asm volatile("" : "+v"(sa), "+v"(sb));
        sa = ok0 ? sa : ninf;
        sb = ok1 ? sb : ninf;
        if (rr == rr0) SSTAMP(4);
        // +-62: the lane's own first maximum (lag j before j + 1), then the wavefront's
        const bool b2 = sb > sa;
        const float na = near0 ? sa : ninf, nb = near1 ? sb : ninf;
        const bool b1 = nb > na;
        float r1, r2; int l1, l2;
#if CWSLG_SEARCH_DPPMAX
        wave_first_max2(b2 ? sb : sa, b2 ? j + 1 : j, b1 ? nb : na, b1 ? j + 1 : j, r2, l2, r1, l1);
#else
        wave_first_max(b2 ? sb : sa, b2 ? j + 1 : j, r2, l2);
        wave_first_max(b1 ? nb : na, b1 ? j + 1 : j, r1, l1);
#endif
        if (lane == 0) {
#if CWSLG_SEARCH_HOISTPTR
            as_global_rw(p_red)[bin] = r1;  as_global_rw(p_jpeak)[bin] = l1;
            as_global_rw(p_red2)[bin] = r2; as_global_rw(p_jpeak2)[bin] = l2;
#else
            as_global_rw(w->red)[bin] = r1;  as_global_rw(w->jpeak)[bin] = l1;       // (HBM addresses: global_store, not flat_store)
            as_global_rw(w->red2)[bin] = r2; as_global_rw(w->jpeak2)[bin] = l2;
#endif
        }
        if (rr == rr0) SSTAMP(5);
        rr = rn;
        live = more;
    }
}

// ft8_sync2d_v3_kernel: one workgroup per (32-bin band, channel), staging as in v2, then sync2d_search_band.  2.74 against v2's 2.92-2.96 ms
// per 4096 slots (same box).  The product runs it (followed by ft8_candidates_kernel) when a boundary carries FEWER channels than the
// chip has workgroup slots for ft8_sync_chan_kernel (two per CU): 29 workgroups per channel fill the chip where one per channel
// walking its 29 bands in turn would not (64 slots: sync stage 0.26 against 0.36 ms).  Same search, same selection: same bits.
__global__ __launch_bounds__(SYNC2D_NT, 4) void ft8_sync2d_v3_kernel(const SyncWork *__restrict__ works, int ia, int ib, int nbins)
{
    constexpr int ROWS = SYNC_BAND + 12, NW = SYNC2D_NT / 64;
    constexpr int UN = (FT8_NHSYM + NW - 1) / NW;                              // symbol steps per wave: 47
    __shared__ __attribute__((aligned(16))) float s_s[ROWS][S2_PITCH];       // s_s[r][m + 2] = s(i0 + r, m)
    __shared__ __attribute__((aligned(16))) float s_c0[NW][S2_PITCH];
    const SyncWork *w = works + blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i0 = ia + blockIdx.x * SYNC_BAND;
    SSTAMP(0);
    {   // staging as in v2: every load of the wave is issued before the first LDS write
        const CWSLG_GLOBAL float *sp = as_global(w->spectra) + i0 + lane;
        const bool on = lane < ROWS && i0 + lane < nbins;
        float v[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int m = wv + NW * q;
            v[q] = (on && m < FT8_NHSYM) ? plane_load1(sp + (size_t)m * nbins) : 0.0f;
        }
        float *dst = &s_s[lane < ROWS ? lane : 0][S2_COL0 + 1 + wv];
#pragma unroll
        for (int q = 0; q < UN; ++q)
            if (lane < ROWS && wv + NW * q < FT8_NHSYM) dst[NW * q] = v[q];
        if (tid < ROWS) {
            float *row = s_s[tid];
            row[0] = 0.0f; row[1] = 0.0f; row[2] = 0.0f;
            row[S2_COL0 + 373] = 0.0f; row[S2_COL0 + 374] = 0.0f; row[S2_COL0 + 375] = 0.0f;
        }
    }
    __syncthreads();
    SSTAMP(1);
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned s_base = (unsigned)(uintptr_t)&s_s[0][0];
    const unsigned sC = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)&s_c0[0][0] + (unsigned)wvu * (S2_PITCH * 4)));
#ifdef CWSLG_SYNC_STAGGER
    if (wvu >= NW / 2) __builtin_amdgcn_s_sleep(CWSLG_SYNC_STAGGER);
#endif
    sync2d_search_band<NW>(w, s_base, sC, wvu, lane, i0, ia, ib);
    SSTAMP(6);
}

// ---------------------------------------------------------------------------------------------
// Bitonic sort of (value, index) keys, ascending, ties by ascending index.  n = 2048, 256 threads.
__device__ __forceinline__ bool key_less(float va, int ia_, float vb, int ib_)
{
    if (va < vb) return true;
    if (va > vb) return false;
    return ia_ < ib_;
}

// n = 1024 or 2048 keys (power of two >= the number of real keys; the padding sorts to the end either way)
template <int NT>
__device__ void bitonic_sort_n(float *kv, int *ki, int n, int tid)
{
    for (int size = 2; size <= n; size <<= 1) {
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            for (int t = tid; t < n / 2; t += NT) {
                const int lo = 2 * t - (t & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const float a = kv[lo], b = kv[hi];
                const int ai = ki[lo], bi = ki[hi];
                const bool sw = up ? key_less(b, bi, a, ai) : key_less(a, ai, b, bi);
                if (sw) { kv[lo] = b; kv[hi] = a; ki[lo] = bi; ki[hi] = ai; }
            }
            __syncthreads();
        }
    }
}

// grid (n_channels), NT threads.  The kernel is a chain of ~165 barrier-separated stages on <= 1024 keys: with 1024 threads every
// stage is one pass (one compare-exchange per thread), so its length is the barrier count, not the key count (round 1 ran it
// with 256 threads: 0.15 ms per 512 channels, now NT = 1024).
// LDS of the candidate selection, carved from one pool so that ft8_sync_chan_kernel can overlay it on its band image.
template <int NT>
struct CandLds {
    static constexpr size_t kv = 0, ki = kv + 2048 * 4, red = ki + 2048 * 4, red2 = red + (FT8_NH1 + 1) * 4 + 12, jp = red2 + (FT8_NH1 + 1) * 4 + 12,
                            jp2 = jp + (FT8_NH1 + 1) * 2 + 14, first = jp2 + (FT8_NH1 + 1) * 2 + 14, second = first + (FT8_NH1 + 2) * 2 + 12,
                            desc = second + (FT8_NH1 + 2) * 2 + 12, scan = desc + SYNC_MAXPRE * 4, scan_lo = scan + NT * 4, cbin = scan_lo + 1024 * 4,
                            clag = cbin + SYNC_MAXPRE * 4, csync = clag + SYNC_MAXPRE * 4, cf = csync + SYNC_MAXPRE * 4, ct = cf + SYNC_MAXPRE * 4,
                            base = ct + SYNC_MAXPRE * 4, n = base + 16, bytes = n + 16;
};

// The candidate selection of one channel by the NT threads of a workgroup: red / jpeak come from global memory (written by this
// workgroup or by an earlier kernel), `lds` holds CandLds<NT>::bytes bytes (16-byte aligned).
template <int NT>
__device__ void ft8_candidates_body(const SyncWork *w, int ia, int ib, float syncmin, int maxcand, int order, char *lds)
{
    using L = CandLds<NT>;
    float *s_kv = reinterpret_cast<float *>(lds + L::kv);
    int *s_ki = reinterpret_cast<int *>(lds + L::ki);
    float *s_red = reinterpret_cast<float *>(lds + L::red), *s_red2 = reinterpret_cast<float *>(lds + L::red2);
    short *s_jp = reinterpret_cast<short *>(lds + L::jp), *s_jp2 = reinterpret_cast<short *>(lds + L::jp2);
    short *s_first = reinterpret_cast<short *>(lds + L::first), *s_second = reinterpret_cast<short *>(lds + L::second);   // bin -> pre-candidate index (or -1)
    int *s_desc = reinterpret_cast<int *>(lds + L::desc);                 // bins in descending red order
    int *s_scan = reinterpret_cast<int *>(lds + L::scan);
    unsigned *s_scan_lo = reinterpret_cast<unsigned *>(lds + L::scan_lo);
    int *s_cbin = reinterpret_cast<int *>(lds + L::cbin), *s_clag = reinterpret_cast<int *>(lds + L::clag);
    float *s_csync = reinterpret_cast<float *>(lds + L::csync), *s_cf = reinterpret_cast<float *>(lds + L::cf), *s_ct = reinterpret_cast<float *>(lds + L::ct);
    float *s_base = reinterpret_cast<float *>(lds + L::base);
    int &s_n = *reinterpret_cast<int *>(lds + L::n);
    const int tid = threadIdx.x;
    const int iz = ib - ia + 1;
    const float df = 12000.0f / 3840.0f, tstep = 480.0f / 12000.0f;
    for (int i = ia + tid; i <= ib; i += NT) {
        s_red[i] = as_global(w->red)[i]; s_red2[i] = as_global(w->red2)[i]; s_jp[i] = (short)as_global(w->jpeak)[i]; s_jp2[i] = (short)as_global(w->jpeak2)[i];
    }
    for (int i = tid; i < FT8_NH1 + 2; i += NT) { s_first[i] = -1; s_second[i] = -1; }
    const int npct = (int)lroundf(0.40f * (float)iz);
    const int lim = min(SYNC_MAXPRE, iz);
    const int nsort = (iz <= 1024) ? 1024 : 2048;       // 200..3000 Hz is 897 bins: the smaller network (55 of 66 stages, half the pairs)
    __syncthreads();
    // --- percentile of red2
    for (int k = tid; k < nsort; k += NT) { s_kv[k] = (k < iz) ? s_red2[ia + k] : __builtin_huge_valf(); s_ki[k] = (k < iz) ? ia + k : 0x7fffffff; }
    __syncthreads();
    bitonic_sort_n<NT>(s_kv, s_ki, nsort, tid);
    if (tid == 0 && npct >= 1) s_base[1] = s_red2[s_ki[npct - 1]];
    __syncthreads();
    // --- order of red (ascending); descending walk list
    for (int k = tid; k < nsort; k += NT) { s_kv[k] = (k < iz) ? s_red[ia + k] : __builtin_huge_valf(); s_ki[k] = (k < iz) ? ia + k : 0x7fffffff; }
    __syncthreads();
    bitonic_sort_n<NT>(s_kv, s_ki, nsort, tid);
    if (tid == 0 && npct >= 1) s_base[0] = s_red[s_ki[npct - 1]];
    for (int r = tid; r < lim; r += NT) s_desc[r] = s_ki[iz - 1 - r];
    __syncthreads();
    if (npct < 1) { if (tid == 0) *as_global_rw(w->ncand) = 0; return; }
    const float base = s_base[0], base2 = s_base[1];
    for (int i = ia + tid; i <= ib; i += NT) { s_red[i] = s_red[i] / base; s_red2[i] = s_red2[i] / base2; }
    __syncthreads();
    // --- walk the bins in descending red; each appends its +-10 peak and, if at another lag, its +-62 peak,
    // until MAXPRECAND entries exist.  Parallel form: per-rank counts -> exclusive scan -> positions < MAXPRECAND.
    constexpr int PER = (SYNC_MAXPRE + NT - 1) / NT;      // ranks per thread
    int cnt[PER], flags[PER];
    int local = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int r = tid * PER + q;
        int f = 0;
        if (r < lim) {
            const int n = s_desc[r];
            if (s_red[n] >= syncmin) f |= 1;
            if (s_jp2[n] != s_jp[n] && s_red2[n] >= syncmin) f |= 2;
        }
        flags[q] = f;
        cnt[q] = (f & 1) + ((f >> 1) & 1);
        local += cnt[q];
    }
    s_scan[tid] = local;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {            // Hillis-Steele inclusive scan
        const int v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int pos = s_scan[tid] - local;                      // exclusive prefix of this thread's first rank
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int r = tid * PER + q;
        if (r < lim) {
            const int n = s_desc[r];
            if ((flags[q] & 1) && pos < SYNC_MAXPRE) { s_cbin[pos] = n; s_clag[pos] = s_jp[n]; s_csync[pos] = s_red[n]; s_first[n] = (short)pos; }
            if (flags[q] & 1) ++pos;
            if ((flags[q] & 2) && pos < SYNC_MAXPRE) { s_cbin[pos] = n; s_clag[pos] = s_jp2[n]; s_csync[pos] = s_red2[n]; s_second[n] = (short)pos; }
            if (flags[q] & 2) ++pos;
        }
    }
    if (tid == NT - 1) s_n = min(s_scan[NT - 1], SYNC_MAXPRE);
    __syncthreads();
    const int ncand = s_n;
    for (int i = tid; i < ncand; i += NT) { s_cf[i] = (float)s_cbin[i] * df; s_ct[i] = ((float)s_clag[i] - 0.5f) * tstep; }
    __syncthreads();
    // --- near-duplicate suppression.  Upstream's sequential double loop
    //       for i: for j<i: if ||f_i|-|f_j|| < 4 Hz and |t_i-t_j| < 0.04 s: zero the weaker of the two (in place)
    // has short-range dependencies only: |df| < 4 Hz <=> bins differ by at most 1 (f = bin*3.125 exactly) and a bin
    // owns at most two pre-candidates, so candidate i meets at most six earlier partners, and its outcome depends
    // only on earlier candidates within +-2 bins.  Evaluate in dependency order: each round, every candidate whose
    // earlier +-2-bin neighbours are all finished runs its inner loop (partners in ascending j, as the original);
    // candidates finished in one round are >= 3 bins apart, so they touch disjoint partners.  Rounds = depth of the
    // dependency chains (a handful for real spectra, ncand in the worst case = the serial loop).
    short *s_done = reinterpret_cast<short *>(s_ki);          // the sort buffers are free now
    short *s_ready = s_done + SYNC_MAXPRE;
    for (int i = tid; i < ncand; i += NT) s_done[i] = 0;
    __syncthreads();
    for (int round = 0; round <= ncand; ++round) {
        int pending = 0;
        for (int i = tid; i < ncand; i += NT) {
            if (s_done[i]) { s_ready[i] = 0; continue; }
            const int n = s_cbin[i];
            bool ok = true;
#pragma unroll
            for (int dn = -2; dn <= 2; ++dn) {
                const int nb = n + dn;
                if (nb < 0 || nb > FT8_NH1) continue;
                const int a = s_first[nb], b = s_second[nb];
                if (a >= 0 && a < i && !s_done[a]) ok = false;
                if (b >= 0 && b < i && !s_done[b]) ok = false;
            }
            s_ready[i] = ok ? 1 : 0;
            pending = 1;
        }
        if (!__syncthreads_or(pending)) break;
        for (int i = tid; i < ncand; i += NT) {
            if (!s_ready[i]) continue;
            const int n = s_cbin[i];
            int part[6];
            int np = 0;
#pragma unroll
            for (int dn = -1; dn <= 1; ++dn) {
                const int nb = n + dn;
                if (nb < 0 || nb > FT8_NH1) continue;
                const int a = s_first[nb], b = s_second[nb];
                if (a >= 0 && a < i) part[np++] = a;
                if (b >= 0 && b < i) part[np++] = b;
            }
#pragma unroll
            for (int x = 1; x < 6; ++x) {                    // insertion sort, ascending j
                if (x < np) {
                    const int v = part[x];
                    int y = x - 1;
                    while (y >= 0 && part[y] > v) { part[y + 1] = part[y]; --y; }
                    part[y + 1] = v;
                }
            }
            const float fi = fabsf(s_cf[i]), ti = s_ct[i];
            float si = s_csync[i];
            for (int x = 0; x < np; ++x) {
                const int j = part[x];
                const float fdiff = fi - fabsf(s_cf[j]);
                const float tdiff = fabsf(ti - s_ct[j]);
                if (fabsf(fdiff) < 4.0f && tdiff < 0.04f) {
                    const float sj = s_csync[j];
                    if (si >= sj) s_csync[j] = 0.0f;
                    if (si < sj) si = 0.0f;
                }
            }
            s_csync[i] = si;
            s_done[i] = 1;
        }
        __syncthreads();
    }
    __syncthreads();
    // --- final order: descending sync, ties ascending bin, then lag (order 0); or ascending bin, entries of one bin in their order of discovery
    // (order 1: sync8.f90's "Sort by frequency" as recalled); first maxcand kept IN THAT ORDER.
    // Bitonic sort of 64-bit keys (inverted order-preserving sync bits | bin | lag | index, or bin | index) in the two sort buffers.
    unsigned *s_hi = reinterpret_cast<unsigned *>(s_kv);
    unsigned *s_lo = reinterpret_cast<unsigned *>(s_scan_lo);
    for (int i = tid; i < 1024; i += NT) {
        unsigned hi = 0xFFFFFFFFu, lo = 0xFFFFFFFFu;
        if (i < ncand && s_csync[i] >= syncmin) {
            unsigned u = __float_as_uint(s_csync[i]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            hi = ~u;                                              // descending sync
            lo = ((unsigned)s_cbin[i] << 20) | ((unsigned)(s_clag[i] + FT8_JZ) << 12) | (unsigned)i;
            if (order) { hi = (unsigned)s_cbin[i]; lo = (unsigned)i; }
        }
        s_hi[i] = hi; s_lo[i] = lo;
    }
    __syncthreads();
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            for (int t = tid; t < 512; t += NT) {
                const int l0 = 2 * t - (t & (stride - 1));
                const int h0 = l0 + stride;
                const bool up = (l0 & size) == 0;
                const unsigned long long ka = ((unsigned long long)s_hi[l0] << 32) | s_lo[l0];
                const unsigned long long kb = ((unsigned long long)s_hi[h0] << 32) | s_lo[h0];
                const bool sw = up ? (kb < ka) : (ka < kb);
                if (sw) { s_hi[l0] = (unsigned)(kb >> 32); s_lo[l0] = (unsigned)kb; s_hi[h0] = (unsigned)(ka >> 32); s_lo[h0] = (unsigned)ka; }
            }
            __syncthreads();
        }
    }
    int nout = 0;
    for (int r = tid; r < 1024; r += NT) {
        if (s_hi[r] == 0xFFFFFFFFu && s_lo[r] == 0xFFFFFFFFu) continue;
        ++nout;
        if (r < maxcand) {
            const int i = (int)(s_lo[r] & 0xFFFu);
            CWSLG_GLOBAL SyncChannelBuffers::Cand *c = as_global_rw(w->cand) + r;
            c->freq_bin = s_cbin[i]; c->time_step = s_clag[i]; c->sync = s_csync[i]; c->freq_hz = s_cf[i]; c->dt_s = s_ct[i];
        }
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (nout) atomicAdd(&s_n, nout);
    __syncthreads();
    if (tid == 0) *as_global_rw(w->ncand) = min(s_n, maxcand);
}

// grid (n_channels), NT threads: the candidate selection as its own launch (behind ft8_sync2d_v3_kernel; ft8_sync_chan_kernel runs the same body itself).
template <int NT>
__global__ __launch_bounds__(NT) void ft8_candidates_kernel(const SyncWork *__restrict__ works, int ia, int ib,
                                                              float syncmin, int maxcand, int order)
{
    __shared__ __attribute__((aligned(16))) char s_pool[CandLds<NT>::bytes];
    ft8_candidates_body<NT>(works + blockIdx.x, ia, ib, syncmin, maxcand, order, s_pool);
}

// ---------------------------------------------------------------------------------------------
// ft8_sync_chan_kernel: the Costas search AND the candidate selection of one channel in one workgroup (grid = channels, 512 threads).
// Same sums in the same order as the three-launch forms (ft8_sync2d_v2 / v3_kernel + ft8_candidates_kernel): bit-identical lists.
// Round-2 form: one workgroup per 32-bin band staged the band's 44 rows as 372 runs of 176 bytes, 3904 bytes apart (each run 2-3
// partial cache lines; neighbouring bands fetched 12 of the 44 rows again: 11.1 GB fetched for a 6.05 GB plane), the workgroup idle
// for a third of its life while they arrived, then a third launch whose ~165 barrier-separated stages left the chip idle for 0.6 ms.
// Here:
//   * the workgroup walks ALL bands of its channel with a sliding LDS window: a band's last 12 rows become the next band's first 12
//     (an LDS move), and only the 32 NEW bins are fetched -- exactly one 128-byte line per symbol step (bands start at bins = 20 mod
//     32 and the spectra rows have a pitch of 32 floats, so bins i0 + 12 .. i0 + 43 are a whole line): every spectrum element leaves
//     HBM once, in whole lines;
//   * the next band's 372 lines are in flight, in 24 registers per lane, while the current band is searched;
//   * the search is the hand-scheduled LDS stream of sync2d_search_band (its temporaries are fixed registers, so the 24 prefetch
//     registers no longer make it spill -- with hipcc's own schedule of the search this kernel was 1.5 % SLOWER than three launches);
//   * when the last band is done the same workgroup runs the candidate selection on the red / jpeak values it has just written
//     (its LDS overlays the band image); other workgroups of the CU are in their search phase meanwhile, so the selection's
//     barrier chains no longer hold the chip.
// Measured (4096 slots, same box, scripts/gpu_r3_sync2d.sh): 3.02 ms against 2.74 + 0.57 (v3 + candidates) and 2.92 + 0.55 (round 2).
constexpr int SYNCC_NT = 512;
__global__ __launch_bounds__(SYNCC_NT, 4) void ft8_sync_chan_kernel(const SyncWork *__restrict__ works, int ia, int ib, int nbins,
                                                                     float syncmin, int maxcand, int order)
{
    constexpr int ROWS = SYNC_BAND + 12, NW = SYNCC_NT / 64;
    constexpr int UN = (FT8_NHSYM + NW - 1) / NW;                              // symbol steps per wave when one step = one load: 47
    constexpr int PF = (FT8_NHSYM + 8 * NW - 1) / (8 * NW);                    // ... 16-byte loads, one wave-level load = eight steps x 128 bytes: 6
    constexpr size_t IMG_BYTES = (size_t)(ROWS + NW) * S2_PITCH * sizeof(float);
    constexpr size_t POOL_BYTES = IMG_BYTES > CandLds<SYNCC_NT>::bytes ? IMG_BYTES : CandLds<SYNCC_NT>::bytes;
    __shared__ __attribute__((aligned(16))) char s_pool[POOL_BYTES];
    float (*s_s)[S2_PITCH] = reinterpret_cast<float (*)[S2_PITCH]>(s_pool);                       // s_s[r][m + 2] = s(i0 + r, m)
    float (*s_c0)[S2_PITCH] = reinterpret_cast<float (*)[S2_PITCH]>(s_pool + (size_t)ROWS * S2_PITCH * sizeof(float));
    const SyncWork *w = works + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // first band: the largest start = 20 (mod 32) that is <= ia (bins below ia are staged but not searched)
    const int i_first = ia - (((ia - 20) % 32) + 32) % 32;
    const int nbands = (ib - i_first) / SYNC_BAND + 1;
    const CWSLG_GLOBAL float *spec = as_global(w->spectra);
    {   // band 0: all 44 rows, one step per load (lanes 0..43), every load of the wave issued before the first LDS write
        const int col = i_first + lane;
        const bool on = lane < ROWS && col >= 0 && col < nbins;
        const CWSLG_GLOBAL float *sp = spec + (on ? col : 0);
        float v[UN];
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const int m = wv + NW * q;
            v[q] = (on && m < FT8_NHSYM) ? plane_load1(sp + (size_t)m * nbins) : 0.0f;
        }
        float *dst = &s_s[lane < ROWS ? lane : 0][S2_COL0 + 1 + wv];
#pragma unroll
        for (int q = 0; q < UN; ++q)
            if (lane < ROWS && wv + NW * q < FT8_NHSYM) dst[NW * q] = v[q];
        if (tid < ROWS) {
            float *row = s_s[tid];
            row[0] = 0.0f; row[1] = 0.0f; row[2] = 0.0f;                        // m = -2, -1, 0
            row[S2_COL0 + 373] = 0.0f; row[S2_COL0 + 374] = 0.0f; row[S2_COL0 + 375] = 0.0f;
        }
    }
    __syncthreads();
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned s_base = (unsigned)(uintptr_t)&s_s[0][0];
    const unsigned sC = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)&s_c0[0][0] + (unsigned)wvu * (S2_PITCH * 4)));
    const int s8 = lane >> 3, g4 = 4 * (lane & 7);         // the prefetch: eight lanes x 16 bytes = the 128-byte line of one step, eight steps per wave-level load
    for (int band = 0; band < nbands; ++band) {
        const int i0 = i_first + band * SYNC_BAND;
        // ---- the NEXT band's 32 new bins (i0 + 44 .. i0 + 75) of every symbol step: one whole 128-byte line per step
        v4f pf[PF];
        const bool more = band + 1 < nbands;               // workgroup-uniform
        if (band == 1) SSTAMP(0);
        if (more) {
            // one uniform base + a 32-bit lane offset per load (a pointer per load would be 12 more registers)
            const int col = i0 + ROWS + g4;
            const bool on = col < nbins;                   // (the row pitch is a multiple of 32 bins: a quad is inside the row or outside it)
            const unsigned o0 = (unsigned)((on ? col : 0) + (8 * wv + s8) * nbins);
            const unsigned ostep = (unsigned)(8 * NW * nbins);
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int m = 8 * (wv + NW * q) + s8;
                pf[q] = (on && m < FT8_NHSYM) ? plane_load4(reinterpret_cast<const CWSLG_GLOBAL v4f *>(spec + o0 + (unsigned)q * ostep)) : v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
        if (band == 1) SSTAMP1(1);
        sync2d_search_band<NW>(w, s_base, sC, wvu, lane, i0, ia, ib);
        if (band == 1) SSTAMP1(6);
        if (!more) break;
        // ---- slide the window: rows 32..43 become rows 0..11 (16 bytes per lane and move), then the prefetched lines fill rows 12..43.
        // The rows are only READ while any wave still searches, so this wave fetches its share of rows 32..43 as soon as its own bins are
        // done; ONE barrier then says both "every wave has finished reading the image" and "rows 32..43 are in registers".
        constexpr int NMV = 12 * S2_PITCH / 4, MVT = (NMV + SYNCC_NT - 1) / SYNCC_NT;       // 1134 float4, 3 per lane
        static_assert((12 * S2_PITCH) % 4 == 0 && (SYNC_BAND * S2_PITCH) % 4 == 0, "16-byte moves");
        v4f mv[MVT];
#pragma unroll
        for (int q = 0; q < MVT; ++q) {
            const int e = tid + SYNCC_NT * q;
            mv[q] = (e < NMV) ? reinterpret_cast<const v4f *>(&s_s[SYNC_BAND][0])[e] : v4f{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < MVT; ++q) {
            const int e = tid + SYNCC_NT * q;
            if (e < NMV) reinterpret_cast<v4f *>(&s_s[0][0])[e] = mv[q];
        }
        {
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int m = 8 * (wv + NW * q) + s8;
                if (m < FT8_NHSYM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s_s[12 + g4 + e][S2_COL0 + 1 + m] = pf[q][e];
                }
            }
        }
        __syncthreads();
        if (band == 1) SSTAMP(7);
    }
    // ---- candidate selection on the values this workgroup has just written (global memory, same CU: visible once the stores have
    // retired, which the barrier's vmcnt(0) ensures)
    __syncthreads();
    ft8_candidates_body<SYNCC_NT>(w, ia, ib, syncmin, maxcand, order, s_pool);
}

// ---------------------------------------------------------------------------------------------
// FT4 candidate search (getcandidates4.f90 + ft4_baseline.f90), one workgroup per channel.  PARITY UNPINNED by the
// reference; bit-exact against oracle/sync_oracle.c, whose builder-defined pieces are mirrored here operation for
// operation: fixed-series double log10 / 10^x, 5x5 normal equations in the scaled variable, Gaussian elimination
// with partial pivoting.
__device__ __forceinline__ double log10_fixed(double x)
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
__device__ __forceinline__ double exp10_fixed(double y)
{
    const double z = y * 3.321928094887362;
    double fl = (double)(long long)z;
    if (fl > z) fl = fl - 1.0;
    const double f = (z - fl) * 0.6931471805599453;
    double s = 1.0;
    for (int k = 22; k >= 1; --k) s = 1.0 + s * f / (double)k;
    long long n = (long long)fl;
    double p = 1.0;
    if (n >= 0) { for (long long q = 0; q < n && q < 2000; ++q) p = p * 2.0; }
    else { for (long long q = 0; q < -n && q < 2000; ++q) p = p * 0.5; }
    return s * p;
}

__global__ __launch_bounds__(256) void ft4_candidates_kernel(const SyncWork *__restrict__ works, int nfa, int nfb,
                                                              float syncmin, int maxcand, int order)
{
    constexpr int NB1 = FT4_NH1 + 1;
    __shared__ float s_savg[NB1], s_savsm[NB1], s_sdb[NB1], s_sbase[NB1];
    __shared__ float s_segbase[10];
    __shared__ double s_a[5];
    __shared__ double s_A[5][6];       // the 5 x 5 system + right-hand side: pivoting indexes its rows dynamically (in registers that is 256 bytes of scratch)
    __shared__ int s_scan[256];
    __shared__ int s_ok;
    __shared__ int s_cbin[SYNC_MAXCAND_CAP];
    __shared__ float s_cs[SYNC_MAXCAND_CAP], s_cf[SYNC_MAXCAND_CAP];
    __shared__ unsigned s_hi[1024], s_lo[1024];
    const SyncWork *w = works + blockIdx.x;
    const int tid = threadIdx.x;
    // the descriptor's pointers are HBM addresses: say so (global_load / global_store instead of flat_*)
    const CWSLG_GLOBAL float *sp = as_global(w->spectra);
    CWSLG_GLOBAL int *ncand_out = as_global_rw(w->ncand);
    const float df = 12000.0f / (float)FT4_NFFT1;
    // averaged spectrum: sum over the 122 symbol steps in order, then /NHSYM
    for (int i = tid; i < NB1; i += 256) {
        float acc = 0.0f;
        if (i >= 1) {
            for (int j = 0; j < FT4_NHSYM; ++j) acc = acc + sp[(size_t)j * FT4_ROW + i];
            acc = acc / (float)FT4_NHSYM;
        }
        s_savg[i] = acc; s_sbase[i] = 0.0f; s_sdb[i] = 0.0f;
    }
    if (tid == 0) s_ok = 1;
    __syncthreads();
    for (int i = tid; i < NB1; i += 256) {
        float t = 0.0f;
        if (i >= 8 && i <= FT4_NH1 - 7) {
#pragma unroll
            for (int q = -7; q <= 7; ++q) t = t + s_savg[i + q];
            t = t / 15.0f;
        }
        s_savsm[i] = t;
    }
    const int ia = nfa, ib = min(nfb, FT4_NH1);
    const int nseg = 10;
    const int nlen = (ib - ia + 1) / nseg, i0 = (ib - ia + 1) / 2;
    const double half = (double)(ib - ia + 1) / 2.0;
    const bool range_ok = (nfb - nfa) >= 20;
    if (!range_ok) { if (tid == 0) *ncand_out = 0; return; }
    for (int i = ia + tid; i <= ib; i += 256) s_sdb[i] = (float)(10.0 * log10_fixed((double)s_savg[i]));
    __syncthreads();
    // 10th percentile of each of the 10 segments (rank of every element inside its segment)
    {
        int jp = (int)lroundf(((float)nlen * 0.01f) * 10.0f);
        if (jp < 1) jp = 1;
        if (jp > nlen) jp = nlen;
        for (int e = tid; e < nseg * nlen; e += 256) {
            const int seg = e / nlen, q = e - seg * nlen;
            const int ja = ia + seg * nlen;
            const float v = s_sdb[ja + q];
            int less = 0, eq = 0;
            for (int x = 0; x < nlen; ++x) {
                const float o = s_sdb[ja + x];
                less += (o < v) ? 1 : 0;
                eq += (o == v) ? 1 : 0;
            }
            if (less <= jp - 1 && jp - 1 < less + eq) s_segbase[seg] = v;
        }
    }
    __syncthreads();
    // lower-envelope points in order -> normal equations -> 5 coefficients (serial: ~100 points, 5x5 system)
    if (tid == 0) {
        double S[9], Tm[5];
        for (int q = 0; q < 9; ++q) S[q] = 0.0;
        for (int q = 0; q < 5; ++q) Tm[q] = 0.0;
        int kz = 0;
        for (int n = 0; n < nseg; ++n) {
            const int ja = ia + n * nlen, jb = ja + nlen - 1;
            const float base = s_segbase[n];
            for (int i = ja; i <= jb; ++i) {
                if (s_sdb[i] <= base && kz < 1000) {
                    ++kz;
                    const double u = (double)(i - i0) / half, y = (double)s_sdb[i];
                    double p = 1.0;
                    for (int q = 0; q < 9; ++q) { S[q] = S[q] + p; if (q < 5) Tm[q] = Tm[q] + y * p; p = p * u; }
                }
            }
        }
        int ok = kz >= 5;
        double (&A)[5][6] = s_A;
        for (int r = 0; r < 5; ++r) { for (int cc = 0; cc < 5; ++cc) A[r][cc] = S[r + cc]; A[r][5] = Tm[r]; }
        for (int col = 0; col < 5 && ok; ++col) {
            int piv = col;
            for (int r = col + 1; r < 5; ++r) if (fabs(A[r][col]) > fabs(A[piv][col])) piv = r;
            if (A[piv][col] == 0.0) { ok = 0; break; }
            if (piv != col) for (int cc = 0; cc < 6; ++cc) { const double t = A[col][cc]; A[col][cc] = A[piv][cc]; A[piv][cc] = t; }
            for (int r = col + 1; r < 5; ++r) {
                const double f = A[r][col] / A[col][col];
                for (int cc = col; cc < 6; ++cc) A[r][cc] = A[r][cc] - f * A[col][cc];
            }
        }
        double a[5] = {0, 0, 0, 0, 0};
        if (ok) {
            for (int r = 4; r >= 0; --r) {
                double t = A[r][5];
                for (int cc = r + 1; cc < 5; ++cc) t = t - A[r][cc] * a[cc];
                a[r] = t / A[r][r];
            }
        }
        for (int q = 0; q < 5; ++q) s_a[q] = a[q];
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) { if (tid == 0) *ncand_out = 0; return; }
    int bad = 0;
    for (int i = ia + tid; i <= ib; i += 256) {
        const double u = (double)(i - i0) / half;
        const float db = (float)(s_a[0] + u * (s_a[1] + u * (s_a[2] + u * (s_a[3] + u * s_a[4]))) + 0.65);
        const float b = (float)exp10_fixed((double)db / 10.0);
        s_sbase[i] = b;
        if (!(b > 0.0f)) bad = 1;
    }
    if (__syncthreads_or(bad)) { if (tid == 0) *ncand_out = 0; return; }
    for (int i = nfa + tid; i <= nfb; i += 256) s_savsm[i] = s_savsm[i] / s_sbase[i];
    __syncthreads();
    {
        CWSLG_GLOBAL float *red = as_global_rw(w->red), *red2 = as_global_rw(w->red2);
        for (int i = tid; i < NB1; i += 256) { red[i] = s_savsm[i]; red2[i] = s_sbase[i]; }   // for parity tests
    }
    // local maxima, ascending bin, first maxcand kept
    constexpr int PER = 5;
    const float f_offset = -1.5f * 12000.0f / 576.0f;
    float spk[PER], fpk[PER];
    int flg[PER];
    int local = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = tid * PER + q;
        flg[q] = 0;
        if (i >= nfa + 1 && i <= nfb - 1) {
            const float c = s_savsm[i], l = s_savsm[i - 1], r = s_savsm[i + 1];
            if (c >= l && c >= r && c >= syncmin) {
                const float den = l - 2.0f * c + r;
                float del = 0.0f;
                if (den != 0.0f) del = 0.5f * (l - r) / den;
                const float fpeak = ((float)i + del) * df + f_offset;
                if (!(fpeak < 200.0f || fpeak > 4910.0f)) {
                    flg[q] = 1;
                    fpk[q] = fpeak;
                    spk[q] = c - 0.25f * (l - r) * del;
                }
            }
        }
        local += flg[q];
    }
    s_scan[tid] = local;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const int v = (tid >= off) ? s_scan[tid - off] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    int pos = s_scan[tid] - local;
    const int keep = min(maxcand, SYNC_MAXCAND_CAP);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (flg[q]) {
            if (pos < keep) { s_cbin[pos] = tid * PER + q; s_cs[pos] = spk[q]; s_cf[pos] = fpk[q]; }
            ++pos;
        }
    }
    const int ncand = min(s_scan[255], keep);
    __syncthreads();
    // descending height, ties by ascending bin (order 0); or as found = ascending bin (order 1)
    for (int i = tid; i < 1024; i += 256) {
        unsigned hi = 0xFFFFFFFFu, lo = 0xFFFFFFFFu;
        if (i < ncand) {
            unsigned u = __float_as_uint(s_cs[i]);
            u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
            hi = ~u;
            lo = ((unsigned)s_cbin[i] << 12) | (unsigned)i;
            if (order) hi = 0u;
        }
        s_hi[i] = hi; s_lo[i] = lo;
    }
    __syncthreads();
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            for (int t = tid; t < 512; t += 256) {
                const int l0 = 2 * t - (t & (stride - 1));
                const int h0 = l0 + stride;
                const bool up = (l0 & size) == 0;
                const unsigned long long ka = ((unsigned long long)s_hi[l0] << 32) | s_lo[l0];
                const unsigned long long kb = ((unsigned long long)s_hi[h0] << 32) | s_lo[h0];
                const bool sw = up ? (kb < ka) : (ka < kb);
                if (sw) { s_hi[l0] = (unsigned)(kb >> 32); s_lo[l0] = (unsigned)kb; s_hi[h0] = (unsigned)(ka >> 32); s_lo[h0] = (unsigned)ka; }
            }
            __syncthreads();
        }
    }
    for (int r = tid; r < ncand; r += 256) {
        const int i = (int)(s_lo[r] & 0xFFFu);
        CWSLG_GLOBAL SyncChannelBuffers::Cand *c = as_global_rw(w->cand) + r;
        c->freq_bin = s_cbin[i]; c->time_step = 0; c->sync = s_cs[i]; c->freq_hz = s_cf[i]; c->dt_s = 0.0f;
    }
    if (tid == 0) *ncand_out = ncand;
}

} // namespace cwslg

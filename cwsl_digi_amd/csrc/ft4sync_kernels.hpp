// ft4sync_kernels.hpp -- FT4's coherent sync stage on gfx950 (the "Costas correlation" of FT4).
//
// *** PARITY UNPINNED by the reference *** (same status as sync_kernels.hpp): CWSL_DIGI spawns jt9 for this.  The kernels
// implement this repository's restatement (oracle/ft4sync_oracle.c) of upstream ft4_decode's candidate refinement --
// ft4_downsample + sync4d + the three-segment coarse/fine search -- operation for operation, so the refined lists are
// BIT-IDENTICAL to that restatement (tests/test_gpu_ft4sync.py).
//
// Per FT4 slot boundary, all FT4 channels batched in each launch:
//   ft4_dft567_kernel        grid (36, channels): the frame's first 72576 samples, packed to 36288 complex = 567 x 64;
//                            567-point DFTs as fmaf chains (4 outputs per lane share every loaded sample), twiddle
//   ft4_fft64_unpack_kernel  grid (284, channels), one wave: rows c and 567-c through 64-point radix-2 DITs in LDS, then
//                            the real-input unpack -> cx[0..36288] (290 KB per channel, lives in L2 / Infinity Cache)
//   ft4_refine_kernel        grid (max_cand, channels): one workgroup per getcandidates4 candidate: the 630 windowed bins
//                            around it -> inverse 4032 = 63 x 64 transform in LDS -> unit-power complex baseband at 666.7 Hz
//                            (kept in LDS as four planes by sample index mod 4: the coarse grid's lanes, 4 samples apart,
//                            then read consecutive addresses) -> sync4d over 3 x (9 x 114 coarse + 9 x 11 fine) grid
//                            points, arg-max by wavefront reduction on an order-preserving key (first maximum in scan order)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sync_kernels.hpp"

namespace cwslg {

constexpr int F4C_NMAX = 72576, F4C_N2 = 36288, F4C_NA = 567, F4C_NP = 4032, F4C_NSS = 32;
constexpr int F4C_KLO = -126, F4C_KHI = 503;
constexpr int F4C_CPT = 4;                      // DFT-567 outputs per lane
constexpr int F4C_PLANE = 1012;                 // 4032 / 4 + pad

struct Ft4Tables {                              // device pointers (built once per context, sync_host.inc)
    const float2 *w567, *wn2, *w2n, *w64, *w63, *w4032, *csync, *ctwk;   // wn2, w4032: [c][b] = W_N^(b c), b < 64 (the other tables: W^k)
    const float *win;
};

struct Ft4Rec { float f0_hz, f1_hz, dt_s, sync; int ibest, idf, seg, cand; };      // == cwslg_ft4_sync

struct alignas(16) Ft4Work {
    const int16_t *frame;
    float2 *y;                                  // [567][64] stage-A output
    float2 *cx;                                 // [36289]
    const SyncChannelBuffers::Cand *cand;
    const int *ncand;
    Ft4Rec *rec;                                // [max_cand][3]
    int *nrec;                                  // [max_cand]
    float2 *cd_dbg;                             // [4032] baseband of candidate 0 (tests)
};

__device__ __forceinline__ float2 cmulc_f(float2 v, float2 w)         // v * conj(w), the restatement's CMULC
{
    return make_float2(__builtin_fmaf(v.x, w.x, v.y * w.y), __builtin_fmaf(v.y, w.x, -(v.x * w.y)));
}
__device__ __forceinline__ int rev6(int b) { return (int)(__brev((unsigned)b) >> 26); }

#if CWSLG_LAB      // the VALU form of the 567-point DFTs (CWSLG_FT4_DFT=valu), lab library only
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ft4_dft567_kernel(const Ft4Work *__restrict__ works, Ft4Tables tb)
{
    __shared__ float2 s_w[F4C_NA];
    const Ft4Work *w = works + blockIdx.y;
    const int tid = threadIdx.x, b = tid & 63;
    for (int k = tid; k < F4C_NA; k += 256) s_w[k] = tb.w567[k];
    __syncthreads();
    const int cbase = __builtin_amdgcn_readfirstlane((int)blockIdx.x * (4 * F4C_CPT) + (tid >> 6) * F4C_CPT);   // wave-uniform
    int idx[F4C_CPT], cc[F4C_CPT];
    float yr[F4C_CPT], yi[F4C_CPT];
#pragma unroll
    for (int i = 0; i < F4C_CPT; ++i) { cc[i] = (cbase + i < F4C_NA) ? cbase + i : 0; idx[i] = 0; yr[i] = 0.f; yi[i] = 0.f; }
    const int *fr = reinterpret_cast<const int *>(w->frame);          // two int16 samples = one packed complex input
#pragma unroll 4
    for (int a = 0; a < F4C_NA; ++a) {
        const int v = fr[64 * a + b];
        const float zr = (float)(short)(v & 0xFFFF), zi = (float)(v >> 16);
#pragma unroll
        for (int i = 0; i < F4C_CPT; ++i) {
            const float2 t = s_w[idx[i]];
            yr[i] = __builtin_fmaf(zr, t.x, yr[i]);
            yr[i] = __builtin_fmaf(-zi, t.y, yr[i]);
            yi[i] = __builtin_fmaf(zr, t.y, yi[i]);
            yi[i] = __builtin_fmaf(zi, t.x, yi[i]);
            idx[i] += cc[i];
            if (idx[i] >= F4C_NA) idx[i] -= F4C_NA;
        }
    }
#pragma unroll
    for (int i = 0; i < F4C_CPT; ++i) {
        const int c = cbase + i;
        if (c < F4C_NA) w->y[c * 64 + rev6(b)] = cmul_f(make_float2(yr[i], yi[i]), tb.wn2[c * 64 + b]);
    }
}
#endif  // CWSLG_LAB

// The same 567-point DFTs on the matrix cores.  Y[c][b] = sum_a W567^(ac) z[a][b] is a dense 567 x 567 by 567 x 64
// complex product, and gfx950's f32 MFMA is bit for bit a k-ordered fmaf chain (cdna_hip_programming.md, 'FP32-input
// MFMA'): v_mfma_f32_32x32x2_f32 with k0 = (wr, zr), k1 = (wi, -zi) performs exactly the restatement's
//   yr = fmaf(zr, wr, yr); yr = fmaf(-zi, wi, yr)        and with k0 = (wi, zr), k1 = (wr, zi)
//   yi = fmaf(zr, wi, yi); yi = fmaf(zi, wr, yi)
// for a 32 (c) x 32 (b) tile per wave, two MFMAs per input row a.  Same bits as ft4_dft567_kernel at a third of the
// time: the VALU version spends its issue slots on the twiddle lookups (one LDS read + index update per term).
typedef float f4c_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void ft4_dft567_mfma_kernel(const Ft4Work *__restrict__ works, Ft4Tables tb)
{
    __shared__ float2 s_w[F4C_NA];
    const Ft4Work *w = works + blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int k = tid; k < F4C_NA; k += 256) s_w[k] = tb.w567[k];
    __syncthreads();
    const int tile = (int)blockIdx.x * 4 + wv;                 // 36 tiles per channel: 18 row tiles x 2 column tiles
    const int c0 = 32 * (tile >> 1), b0 = 32 * (tile & 1);     // (nine workgroups per channel: every SIMD gets the same load)
    const int i = lane & 31, h = lane >> 5;
    const int c = (c0 + i < F4C_NA) ? c0 + i : 0;             // rows past 566 compute row 0 and are not stored
    const int b = b0 + i;
    const CWSLG_GLOBAL int *fr = as_global(reinterpret_cast<const int *>(w->frame)) + b;   // two int16 = one packed complex input
    f4c_f32x16 yr, yi;
#pragma unroll
    for (int v = 0; v < 16; ++v) { yr[v] = 0.f; yi[v] = 0.f; }
    int idx = 0;
#pragma unroll 4
    for (int a = 0; a < F4C_NA; ++a) {
        const int zv = fr[64 * a];
        const float zr = (float)(short)(zv & 0xFFFF), zi = (float)(zv >> 16);
        const float2 t = s_w[idx];
        idx += c;
        if (idx >= F4C_NA) idx -= F4C_NA;
        // lane (i, h): A[i][k = h], B[k = h][j = i]
        yr = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? t.y : t.x, h ? -zi : zr, yr, 0, 0, 0);
        yi = __builtin_amdgcn_mfma_f32_32x32x2f32(h ? t.x : t.y, h ? zi : zr, yi, 0, 0, 0);
    }
    // D[row][col]: col = lane & 31 (b), row = (v & 3) + 8 (v >> 2) + 4 (lane >> 5) (c)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int cr = c0 + (v & 3) + 8 * (v >> 2) + 4 * h;
        if (cr < F4C_NA) gst2(w->y + cr * 64 + rev6(b), cmul_f(make_float2(yr[v], yi[v]), tb.wn2[cr * 64 + b]));
    }
}

// one 64-point radix-2 DIT stage on two rows held by one wave: lane -> (row lane >> 5, butterfly lane & 31)
template <bool INVERSE>
__device__ __forceinline__ void fft64_stage(float2 *r0, float2 *r1, const float2 *s_w64, int len, int lane)
{
    float2 *row = (lane & 32) ? r1 : r0;
    const int j = lane & 31, half = len >> 1;
    const int k = j & (half - 1), base = (j / half) * len;
    const float2 u = row[base + k], v = row[base + k + half];
    const float2 wv = s_w64[k * (64 / len)];
    const float2 t = INVERSE ? cmulc_f(v, wv) : cmul_f(v, wv);
    row[base + k] = make_float2(u.x + t.x, u.y + t.y);
    row[base + k + half] = make_float2(u.x - t.x, u.y - t.y);
}
__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float2 unpack_bin(float2 A, float2 Bc, float2 wk)      // Bc = conj(Z[N-k]) already
{
    const float er = (A.x + Bc.x) * 0.5f, ei = (A.y + Bc.y) * 0.5f;
    const float2 o = make_float2((A.x - Bc.x) * 0.5f, (A.y - Bc.y) * 0.5f);
    const float2 t = cmul_f(o, wk);
    return make_float2(er + t.y, ei - t.x);
}

__global__ __launch_bounds__(64) void ft4_fft64_unpack_kernel(const Ft4Work *__restrict__ works, Ft4Tables tb)
{
    __shared__ float2 s_r[2][64];
    __shared__ float2 s_w64[32];
    const Ft4Work *w = works + blockIdx.y;
    const int p = blockIdx.x, lane = threadIdx.x;             // rows p and 567 - p (p = 0: row 0 alone)
    const int c0 = p, c1 = (p == 0) ? 0 : F4C_NA - p;
    s_r[0][lane] = gld2(w->y + c0 * 64 + lane);
    s_r[1][lane] = gld2(w->y + c1 * 64 + lane);
    if (lane < 32) s_w64[lane] = tb.w64[lane];
    wave_sync_lds();
    for (int len = 2; len <= 64; len <<= 1) {
        fft64_stage<false>(s_r[0], s_r[1], s_w64, len, lane);
        wave_sync_lds();
    }
    // Z[c + 567 d] = row_c[d];  N2 - (c + 567 d) = (567 - c) + 567 (63 - d)   (c > 0)
    const int d = lane;
    float2 *cx = w->cx;          // (stored through gst2: HBM address)
    if (p > 0) {
        const int k0 = c0 + F4C_NA * d, k1 = c1 + F4C_NA * d;
        float2 B = s_r[1][63 - d]; B.y = -B.y;
        gst2(cx + (k0), unpack_bin(s_r[0][d], B, tb.w2n[k0]));
        B = s_r[0][63 - d]; B.y = -B.y;
        gst2(cx + (k1), unpack_bin(s_r[1][d], B, tb.w2n[k1]));
    } else {
        const int k0 = F4C_NA * d;
        float2 B = s_r[0][(64 - d) & 63]; B.y = -B.y;         // d = 0 pairs with itself
        gst2(cx + (k0), unpack_bin(s_r[0][d], B, tb.w2n[k0]));
        if (d == 0) {                                         // k = N2: Z[0] with conj(Z[0])
            float2 B0 = s_r[0][0]; B0.y = -B0.y;
            gst2(cx + (F4C_N2), unpack_bin(s_r[0][0], B0, tb.w2n[F4C_N2]));
        }
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long f4_key(float v, unsigned order)
{
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);              // total order of floats as unsigned
    return ((unsigned long long)u << 32) | (0xFFFFFFFFu - order); // ties: the earlier grid point wins ("sync > smax")
}

struct F4Cd {                                                     // the baseband, four planes by sample index mod 4
    const float2 (*pl)[F4C_PLANE];
    __device__ __forceinline__ float2 at(int m) const { return pl[m & 3][m >> 2]; }
};

// one Costas block: terms ka..kb-1 of the tweaked reference, data from sample `start`, every other sample
__device__ __forceinline__ float2 f4_corr(const F4Cd &cd, int start, const float2 *cs, int ka, int kb)
{
    float zr = 0.f, zi = 0.f;
    for (int k = ka; k < kb; ++k) {
        const float2 c = cd.at(start + 2 * (k - ka)), s = cs[k];
        zr = __builtin_fmaf(c.x, s.x, zr); zr = __builtin_fmaf(c.y, s.y, zr);
        zi = __builtin_fmaf(c.y, s.x, zi); zi = __builtin_fmaf(-c.x, s.y, zi);
    }
    return make_float2(zr, zi);
}
__device__ __forceinline__ float2 f4_corr_full(const F4Cd &cd, int start, const float2 *cs)
{
    float zr = 0.f, zi = 0.f;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        const float2 c = cd.at(start + 2 * k), s = cs[k];
        zr = __builtin_fmaf(c.x, s.x, zr); zr = __builtin_fmaf(c.y, s.y, zr);
        zi = __builtin_fmaf(c.y, s.x, zi); zi = __builtin_fmaf(-c.x, s.y, zi);
    }
    return make_float2(zr, zi);
}
__device__ __forceinline__ float f4_pmag(float2 z)
{
    const float fac = 1.0f / 64.0f;
    const float a = z.x * fac, b = z.y * fac;
    return sqrtf(__builtin_fmaf(a, a, b * b));
}
// sync4d(cd, i0, ctwk(:, idf)): cs = the four tweaked reference blocks [4][64]
__device__ __forceinline__ float f4_sync4d(const F4Cd &cd, int i0, const float2 *cs)
{
    const int i1 = i0, i2 = i0 + 33 * F4C_NSS, i3 = i0 + 66 * F4C_NSS, i4 = i0 + 99 * F4C_NSS;
    const int last = 4 * F4C_NSS - 1;
    float2 z1 = make_float2(0.f, 0.f), z2 = z1, z3 = z1, z4 = z1;
    if (i1 >= 0 && i1 + last <= F4C_NP - 1) z1 = f4_corr_full(cd, i1, cs);
    if (i1 < 0) {
        const int npts = (i1 + last) / 2;
        if (npts > 16) z1 = f4_corr(cd, 0, cs, 63 - npts, 64);
    }
    if (i2 >= 0 && i2 + last <= F4C_NP - 1) z2 = f4_corr_full(cd, i2, cs + 64);
    if (i3 >= 0 && i3 + last <= F4C_NP - 1) z3 = f4_corr_full(cd, i3, cs + 128);
    if (i4 >= 0 && i4 + last <= F4C_NP - 1) z4 = f4_corr_full(cd, i4, cs + 192);
    if (i4 + last > F4C_NP - 1) {
        const int npts = (F4C_NP - 1 - i4 + 1) / 2;
        z4 = (npts > 16) ? f4_corr(cd, i4, cs + 192, 0, npts) : make_float2(0.f, 0.f);
    }
    return ((f4_pmag(z1) + f4_pmag(z2)) + f4_pmag(z3)) + f4_pmag(z4);
}

__global__ __launch_bounds__(256) void ft4_refine_kernel(const Ft4Work *__restrict__ works, Ft4Tables tb, int max_cand)
{
    __shared__ float2 s_y[63][64];                 // stage-A output / FFT rows; afterwards the 9 tweaked reference sets
    __shared__ float2 s_cd[4][F4C_PLANE];
    __shared__ float2 s_c1[10][64];
    __shared__ float2 s_w63[63];
    __shared__ float2 s_w64[32];
    __shared__ float s_part[256];
    __shared__ unsigned long long s_key[4];
    const Ft4Work *w = works + blockIdx.y;
    const int cand = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int ncand = *as_global(w->ncand);
    if (ncand > max_cand) ncand = max_cand;
    if (cand >= ncand) return;                      // workgroup-uniform
    const float f0 = as_global(w->cand)[cand].freq_hz;
    const float df = 12000.0f / (float)F4C_NMAX;
    const int i0 = (int)lroundf(f0 / df);

    // ---- the live rows of the inverse transform's input: j = 64 a + b, a in {0..7, 61, 62}
    for (int e = tid; e < 640; e += 256) {
        const int q = e >> 6, b = e & 63;
        const int a = (q < 8) ? q : 53 + q;          // q = 8, 9 -> a = 61, 62
        const int j = 64 * a + b;
        const int k = (q < 8) ? j : j - F4C_NP;
        float2 v = make_float2(0.f, 0.f);
        const int idx = i0 + k;
        if (k >= F4C_KLO && k <= F4C_KHI && idx >= 0 && idx <= F4C_N2) {
            const float2 x = gld2(w->cx + idx);
            const float wk = tb.win[k - F4C_KLO];
            v = make_float2((x.x * wk) / 4032.0f, (x.y * wk) / 4032.0f);
        }
        s_c1[q][b] = v;
    }
    if (tid < 63) s_w63[tid] = tb.w63[tid];
    if (tid >= 64 && tid < 96) s_w64[tid - 64] = tb.w64[tid - 64];
    __syncthreads();
    // ---- stage A: 63-point inverse DFT over the 10 live rows, twiddle conj(W4032^(bc)), bit-reversed store
    {
        const int b = lane;
        for (int c = wv; c < 63; c += 4) {
            float yr = 0.f, yi = 0.f;
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const int a = (q < 8) ? q : 53 + q;
                const float2 z = s_c1[q][b], t = s_w63[(a * c) % 63];
                yr = __builtin_fmaf(z.x, t.x, yr); yr = __builtin_fmaf(z.y, t.y, yr);
                yi = __builtin_fmaf(z.y, t.x, yi); yi = __builtin_fmaf(-z.x, t.y, yi);
            }
            s_y[c][rev6(b)] = cmulc_f(make_float2(yr, yi), tb.w4032[c * 64 + b]);
        }
    }
    wave_sync_lds();                                 // rows c = wv, wv+4, ... were written by this wave only
    // ---- 64-point inverse DITs: the wave's own rows, two at a time
    for (int len = 2; len <= 64; len <<= 1) {
        for (int i = 0; i < 8; ++i) {
            const int ra = wv + 8 * i, rb = wv + 8 * i + 4;     // rows of lanes 0-31 / 32-63
            if (rb < 63) fft64_stage<true>(s_y[ra], s_y[rb], s_w64, len, lane);
            else if (ra < 63 && lane < 32) fft64_stage<true>(s_y[ra], s_y[ra], s_w64, len, lane);
        }
        wave_sync_lds();
    }
    __syncthreads();
    // ---- cd[m] = y[m % 63][m / 63]; mean power by 256 strided partial sums and a halving tree; normalise
    {
        float s = 0.f;
        for (int m = tid; m < F4C_NP; m += 256) {
            const float2 v = s_y[m % 63][m / 63];
            s_cd[m & 3][m >> 2] = v;
            s = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, s));
        }
        s_part[tid] = s;
        __syncthreads();
        for (int h = 128; h >= 1; h >>= 1) {
            if (tid < h) s_part[tid] = s_part[tid] + s_part[tid + h];
            __syncthreads();
        }
        const float sum2 = s_part[0] / 4032.0f;
        if (sum2 > 0.0f) {
            const float sc = sqrtf(sum2);
            for (int m = tid; m < F4C_NP; m += 256) {
                float2 v = s_cd[m & 3][m >> 2];
                v.x = v.x / sc; v.y = v.y / sc;
                s_cd[m & 3][m >> 2] = v;
            }
        }
        __syncthreads();
        if (cand == 0 && w->cd_dbg)
            for (int m = tid; m < F4C_NP; m += 256) gst2(w->cd_dbg + m, s_cd[m & 3][m >> 2]);
    }
    // ---- the search of ft4_decode: 3 segments x (coarse, fine)
    F4Cd cd{s_cd};
    float2 *s_cs = &s_y[0][0];                       // [9][256] tweaked references (s_y is free now)
    float smax = -99.0f, smax1 = 0.0f;
    int nrec = 0;
    for (int iseg = 1; iseg <= 3; ++iseg) {
        int ibest = -1, idfbest = 0;
        for (int isync = 1; isync <= 2; ++isync) {
            int idfmin, idfstp, ibmin, ibmax, ibstp;
            if (isync == 1) {
                idfmin = -12; idfstp = 3; ibstp = 4;
                if (iseg == 1) { ibmin = 108; ibmax = 560; }
                else if (iseg == 2) { smax1 = smax; ibmin = 560; ibmax = 1012; }
                else { ibmin = -344; ibmax = 108; }
            } else {
                idfmin = idfbest - 4; idfstp = 1;
                ibmin = ibest - 5; ibmax = ibest + 5; ibstp = 1;
            }
            const int nib = (ibmax - ibmin) / ibstp + 1;
            __syncthreads();                             // previous grid's readers are done with s_cs
            for (int e = tid; e < 9 * 256; e += 256) {
                const int dd = e >> 8, r = e & 255;
                const int idf = idfmin + dd * idfstp;
                s_cs[e] = cmul_f(tb.ctwk[(idf + 16) * 64 + (r & 63)], tb.csync[r]);
            }
            __syncthreads();
            unsigned long long key = 0ull;
            for (int pt = tid; pt < 9 * nib; pt += 256) {
                const int dd = pt / nib, ii = pt - dd * nib;
                const float sy = f4_sync4d(cd, ibmin + ii * ibstp, s_cs + 256 * dd);
                const unsigned long long k = f4_key(sy, (unsigned)pt);
                key = (k > key) ? k : key;
            }
            key = wave_max_u64(key);
            if (lane == 0) s_key[wv] = key;
            __syncthreads();
            unsigned long long best = s_key[0];
#pragma unroll
            for (int q = 1; q < 4; ++q) best = (s_key[q] > best) ? s_key[q] : best;
            const int pt = (int)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
            const int dd = pt / nib, ii = pt - dd * nib;
            smax = key_value(best);
            ibest = ibmin + ii * ibstp;
            idfbest = idfmin + dd * idfstp;
        }
        if (iseg == 1) smax1 = smax;
        if (smax < 1.2f) continue;
        if (iseg > 1 && smax < smax1) continue;
        const float f1 = f0 + (float)idfbest;
        if (f1 <= 10.0f || f1 >= 4990.0f) continue;
        if (tid == 0) {
            CWSLG_GLOBAL Ft4Rec *r = as_global_rw(w->rec) + (cand * 3 + nrec);
            r->f0_hz = f0; r->f1_hz = f1; r->dt_s = (float)ibest / 666.67f - 0.5f; r->sync = smax;
            r->ibest = ibest; r->idf = idfbest; r->seg = iseg; r->cand = cand;
        }
        ++nrec;
    }
    if (tid == 0) as_global_rw(w->nrec)[cand] = nrec;
}

}  // namespace cwslg

// sync_kernels.hpp -- FT8 symbol spectra + Costas-array sync search on gfx950.
//
// *** PARITY UNPINNED by the reference ***: CWSL_DIGI has no sync code; it spawns WSJT-X jt9.exe
// (source/DecoderPool.hpp:634-676), which is not vendored, not version-pinned and not in the build
// container.  These kernels implement the published FT8 candidate search (WSJT-X 2.6.x lib/ft8/sync8.f90,
// ft8_params.f90: NSPS=1920, NFFT1=3840, NSTEP=480, NHSYM=372, JZ=62, icos7 = 3,1,4,0,6,5,2) with an
// arithmetic specification shared with the repository's own CPU restatement (oracle/sync_oracle.c):
// every float operation is un-fused and in a fixed order, the FFT factorisation and twiddle tables are
// fixed, so the candidate lists are BIT-IDENTICAL to that restatement (tests/test_gpu_sync.py).
//
// Three launches per slot boundary, all channels of the group batched in each:
//   ft8_spectra_kernel    one workgroup per (symbol step, channel): 1920 int16 -> packed 1920-point complex
//                         FFT (15 x 128: 15-point DFTs + radix-2 DIT butterflies in LDS) -> |X|^2 rows
//   ft8_sync2d_kernel     one workgroup per (32-bin band, channel): band of the spectra staged in LDS once,
//                         lane = time lag, Costas correlation for 125 lags, wavefront-shuffle arg-max for the
//                         +-10 and +-62 lag peak searches
//   ft8_candidates_kernel one workgroup per channel: 40th-percentile normalisation (bitonic sort in LDS),
//                         thresholding, near-duplicate suppression, final ordering
// Why not one fused kernel: one slot's spectra are 372 x ~973 floats = 1.45 MB, nine times the CU's LDS, and
// every lag touches 21 symbol steps spread over the whole slot; the spectra make one trip through
// L2/Infinity Cache between the first two launches instead (1.45 MB per slot against 23 MB of IQ).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cwslg {

constexpr int FT8_NSPS = 1920, FT8_NSTEP = 480, FT8_NHSYM = 372, FT8_NH1 = 1920, FT8_JZ = 62, FT8_NMAX = 180000;
constexpr int SYNC_BAND = 32;           // bins per sync2d workgroup
constexpr int SYNC_MAXCAND_CAP = 600;   // MAXCAND of the upstream decoder

struct SyncConfig {
    bool enabled = false;
    float syncmin = 1.5f;
    int max_cand = 200;
    int f_lo_hz = 200, f_hi_hz = 3000;
    int ia = 64, ib = 960, nbins = 976;   // derived: bin range and stored row length (ib+13 rounded up to 16)
};

struct SyncTables {                       // device pointers
    const float2 *w15, *w1920, *w128, *w3840;
};

struct SyncShared {
    float2 *d_tables = nullptr;           // w15[15] | w1920[1920] | w128[64] | w3840[1921]
    SyncTables t{};
};

struct SyncChannelBuffers {
    char *d_block = nullptr;
    float *d_spectra = nullptr;           // [NHSYM][nbins]
    float *d_red = nullptr, *d_red2 = nullptr;      // [NH1+1]
    int *d_jpeak = nullptr, *d_jpeak2 = nullptr;    // [NH1+1]
    struct Cand { int freq_bin, time_step; float sync, freq_hz, dt_s; } *d_cand = nullptr;   // [max_cand]
    int *d_ncand = nullptr;
    int nbins = 0, max_cand = 0;
};

struct alignas(16) SyncWork {
    const int16_t *frame;
    float *spectra;
    float *red, *red2;
    int *jpeak, *jpeak2;
    SyncChannelBuffers::Cand *cand;
    int *ncand;
};

inline void sync_free_channel(SyncChannelBuffers &b)
{
    if (b.d_block) (void)hipFree(b.d_block);
    b = SyncChannelBuffers();
}
inline void sync_free_shared(SyncShared &s)
{
    if (s.d_tables) (void)hipFree(s.d_tables);
    s = SyncShared();
}

__device__ __forceinline__ float2 cmul_u(float2 a, float2 b)          // un-fused complex product
{
    const float ac = a.x * b.x, bd = a.y * b.y, ad = a.x * b.y, bc = a.y * b.x;
    return make_float2(ac - bd, ad + bc);
}

// ---------------------------------------------------------------------------------------------
// grid (NHSYM, n_channels), 256 threads.
__global__ __launch_bounds__(256) void ft8_spectra_kernel(const SyncWork *__restrict__ works, SyncTables tb, int nbins)
{
    __shared__ float s_x[FT8_NSPS];
    __shared__ float2 s_y[15][128];
    const SyncWork *w = works + blockIdx.y;
    const int j = blockIdx.x;
    const int tid = threadIdx.x;
    const int16_t *d = w->frame + (size_t)FT8_NSTEP * j;
    const float fac = 1.0f / 300.0f;
    for (int n = tid; n < FT8_NSPS; n += 256) s_x[n] = fac * (float)d[n];
    __syncthreads();

    // stage 1: 15-point DFT over a (z[128a+b], a<=7 non-zero), twiddle W1920^(bc), store bit-reversed in b
    {
        const int b = tid & 127;
        const int chalf = __builtin_amdgcn_readfirstlane(tid >> 7);     // wave-uniform: waves 0,1 -> 0 ; 2,3 -> 1
        float2 z[8];
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            const int m = 128 * a + b;
            z[a] = (m < 960) ? make_float2(s_x[2 * m], s_x[2 * m + 1]) : make_float2(0.f, 0.f);
        }
        const int rb = (int)(__brev((unsigned)b) >> 25);
        for (int c = chalf; c < 15; c += 2) {
            float2 acc = z[0];
#pragma unroll
            for (int a = 1; a < 8; ++a) {
                const int m = 128 * a + b;
                if (m < 960) {
                    const float2 p = cmul_u(z[a], tb.w15[(a * c) % 15]);
                    acc.x = acc.x + p.x;
                    acc.y = acc.y + p.y;
                }
            }
            s_y[c][rb] = cmul_u(acc, tb.w1920[b * c]);
        }
    }
    __syncthreads();

    // stage 2: 15 radix-2 DIT FFTs of 128 points, 960 butterflies per stage
#pragma unroll 1
    for (int len = 2; len <= 128; len <<= 1) {
        const int half = len >> 1, step = 128 / len;
        for (int idx = tid; idx < 960; idx += 256) {
            const int c = idx >> 6, q = idx & 63;
            const int k = q & (half - 1);
            const int i0 = ((q / half) * len) + k, i1 = i0 + half;
            const float2 u = s_y[c][i0], v = s_y[c][i1];
            const float2 t = cmul_u(v, tb.w128[k * step]);
            s_y[c][i0] = make_float2(u.x + t.x, u.y + t.y);
            s_y[c][i1] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }

    // stage 3: unpack the real-input transform, power spectrum
    float *out = w->spectra + (size_t)j * nbins;
    for (int k = tid; k < nbins; k += 256) {
        float pw = 0.0f;
        if (k <= FT8_NH1) {
            const int k2 = (1920 - k) % 1920, kk = k % 1920;
            const float2 A = s_y[kk % 15][kk / 15];
            float2 B = s_y[k2 % 15][k2 / 15];
            B.y = -B.y;
            const float er = (A.x + B.x) * 0.5f, ei = (A.y + B.y) * 0.5f;
            const float2 o = make_float2((A.x - B.x) * 0.5f, (A.y - B.y) * 0.5f);
            const float2 t = cmul_u(o, tb.w3840[k]);
            const float xr = er + t.y, xi = ei - t.x;
            pw = xr * xr + xi * xi;
        }
        out[k] = pw;
    }
}

// ---------------------------------------------------------------------------------------------
// grid (ceil((ib-ia+1)/32), n_channels), 256 threads.  LDS: band rows [44][376] + 7-tone sums [32][376].
__global__ __launch_bounds__(256) void ft8_sync2d_kernel(const SyncWork *__restrict__ works, int ia, int ib, int nbins)
{
    constexpr int ROWS = SYNC_BAND + 12, PITCH = 376;
    __shared__ float s_s[ROWS][PITCH];
    __shared__ float s_c0[SYNC_BAND][PITCH];
    __shared__ float s_rv[4][2];
    __shared__ int s_rj[4][2];
    const SyncWork *w = works + blockIdx.y;
    const int i0 = ia + blockIdx.x * SYNC_BAND;
    const int tid = threadIdx.x;
    // stage the band: s_s[r][m-1] = s(i0+r, m)
    for (int e = tid; e < ROWS * FT8_NHSYM; e += 256) {
        const int m = e / ROWS, r = e - m * ROWS;
        const int bin = i0 + r;
        s_s[r][m] = (bin < nbins) ? w->spectra[(size_t)m * nbins + bin] : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < SYNC_BAND * FT8_NHSYM; e += 256) {
        const int r = e / FT8_NHSYM, m = e - r * FT8_NHSYM;
        float c0 = 0.0f;
#pragma unroll
        for (int k = 0; k < 7; ++k) c0 = c0 + s_s[r + 2 * k][m];
        s_c0[r][m] = c0;
    }
    __syncthreads();

    const int lane = tid & 63, wv = tid >> 6;
    const int l = (wv & 1) * 64 + lane;             // lag index 0..127 ; j = l - 62
    const int j = l - FT8_JZ;
    const bool lag_ok = l <= 2 * FT8_JZ;
    const int icos[7] = {3, 1, 4, 0, 6, 5, 2};
    for (int rr = wv >> 1; rr < SYNC_BAND; rr += 2) {          // two bins per iteration (wave pairs)
        const int bin = i0 + rr;
        float sy = 0.0f;
        if (lag_ok && bin <= ib) {
            float ta = 0, tbv = 0, tc = 0, t0a = 0, t0b = 0, t0c = 0;
#pragma unroll
            for (int n = 0; n < 7; ++n) {
                const int m = j + 12 + 4 * n;                  // 1-based symbol-step index
                const int row = rr + 2 * icos[n];
                if (m >= 1 && m <= FT8_NHSYM) { ta = ta + s_s[row][m - 1]; t0a = t0a + s_c0[rr][m - 1]; }
                { const int mb = m + 144; tbv = tbv + s_s[row][mb - 1]; t0b = t0b + s_c0[rr][mb - 1]; }
                if (m + 288 <= FT8_NHSYM) { const int mc = m + 288; tc = tc + s_s[row][mc - 1]; t0c = t0c + s_c0[rr][mc - 1]; }
            }
            float t = ta + tbv + tc;
            float t0 = t0a + t0b + t0c;
            t0 = (t0 - t) / 6.0f;
            const float sync_abc = t / t0;
            t = tbv + tc;
            t0 = t0b + t0c;
            t0 = (t0 - t) / 6.0f;
            const float sync_bc = t / t0;
            sy = (sync_abc > sync_bc) ? sync_abc : sync_bc;
            if (!(sy == sy)) sy = 0.0f;                        // 0/0 on all-zero windows: defined as 0 (as the oracle)
        }
        // wavefront arg-max (first maximum wins, like maxloc) for the two searches
        float v2 = sy; int j2 = j; bool h2 = lag_ok;
        float v1 = sy; int j1 = j; bool h1 = lag_ok && j >= -10 && j <= 10;
#pragma unroll
        for (int msk = 32; msk >= 1; msk >>= 1) {
            {
                const float ov = __shfl_xor(v2, msk, 64); const int oj = __shfl_xor(j2, msk, 64); const int oh = __shfl_xor((int)h2, msk, 64);
                if (oh && (!h2 || ov > v2 || (ov == v2 && oj < j2))) { v2 = ov; j2 = oj; h2 = true; }
            }
            {
                const float ov = __shfl_xor(v1, msk, 64); const int oj = __shfl_xor(j1, msk, 64); const int oh = __shfl_xor((int)h1, msk, 64);
                if (oh && (!h1 || ov > v1 || (ov == v1 && oj < j1))) { v1 = ov; j1 = oj; h1 = true; }
            }
        }
        if (lane == 0) { s_rv[wv][0] = v1; s_rj[wv][0] = h1 ? j1 : 9999; s_rv[wv][1] = v2; s_rj[wv][1] = h2 ? j2 : 9999; }
        __syncthreads();
        if (lane == 0 && (wv & 1) == 0 && bin <= ib) {
            // combine the two waves of this bin (lags -62..1 | 2..62); sequential first-maximum rule
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                float va = s_rv[wv][which], vb = s_rv[wv + 1][which];
                int ja = s_rj[wv][which], jb = s_rj[wv + 1][which];
                float best; int bj;
                if (ja == 9999) { best = vb; bj = jb; }
                else if (jb == 9999) { best = va; bj = ja; }
                else if (vb > va) { best = vb; bj = jb; }
                else { best = va; bj = ja; }
                if (which == 0) { w->red[bin] = best; w->jpeak[bin] = bj; }
                else { w->red2[bin] = best; w->jpeak2[bin] = bj; }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Bitonic sort of (value, index) keys, ascending, ties by ascending index.  n = 2048, 256 threads.
__device__ __forceinline__ bool key_less(float va, int ia_, float vb, int ib_)
{
    if (va < vb) return true;
    if (va > vb) return false;
    return ia_ < ib_;
}

__device__ void bitonic_sort_2048(float *kv, int *ki, int tid)
{
    for (int size = 2; size <= 2048; size <<= 1) {
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            for (int t = tid; t < 1024; t += 256) {
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

// grid (n_channels), 256 threads.
__global__ __launch_bounds__(256) void ft8_candidates_kernel(const SyncWork *__restrict__ works, int ia, int ib,
                                                              float syncmin, int maxcand)
{
    __shared__ float s_kv[2048];
    __shared__ int s_ki[2048];
    __shared__ float s_red[FT8_NH1 + 1], s_red2[FT8_NH1 + 1];
    __shared__ short s_jp[FT8_NH1 + 1], s_jp2[FT8_NH1 + 1];
    __shared__ int s_desc[SYNC_MAXCAND_CAP];            // bins in descending red order
    __shared__ int s_cnt[2];
    __shared__ int s_cbin[SYNC_MAXCAND_CAP], s_clag[SYNC_MAXCAND_CAP];
    __shared__ float s_csync[SYNC_MAXCAND_CAP], s_cf[SYNC_MAXCAND_CAP], s_ct[SYNC_MAXCAND_CAP];
    __shared__ float s_base[2];
    __shared__ int s_n;
    const SyncWork *w = works + blockIdx.x;
    const int tid = threadIdx.x;
    const int iz = ib - ia + 1;
    const float df = 12000.0f / 3840.0f, tstep = 480.0f / 12000.0f;
    for (int i = ia + tid; i <= ib; i += 256) {
        s_red[i] = w->red[i]; s_red2[i] = w->red2[i]; s_jp[i] = (short)w->jpeak[i]; s_jp2[i] = (short)w->jpeak2[i];
    }
    const int npct = (int)lroundf(0.40f * (float)iz);
    const int lim = min(min(maxcand, iz), SYNC_MAXCAND_CAP);
    __syncthreads();
    // --- percentile of red2
    for (int k = tid; k < 2048; k += 256) { s_kv[k] = (k < iz) ? s_red2[ia + k] : __builtin_huge_valf(); s_ki[k] = (k < iz) ? ia + k : 0x7fffffff; }
    __syncthreads();
    bitonic_sort_2048(s_kv, s_ki, tid);
    if (tid == 0 && npct >= 1) s_base[1] = s_red2[s_ki[npct - 1]];
    __syncthreads();
    // --- order of red (ascending); descending walk list
    for (int k = tid; k < 2048; k += 256) { s_kv[k] = (k < iz) ? s_red[ia + k] : __builtin_huge_valf(); s_ki[k] = (k < iz) ? ia + k : 0x7fffffff; }
    __syncthreads();
    bitonic_sort_2048(s_kv, s_ki, tid);
    if (tid == 0 && npct >= 1) s_base[0] = s_red[s_ki[npct - 1]];
    for (int r = tid; r < lim; r += 256) s_desc[r] = s_ki[iz - 1 - r];
    __syncthreads();
    if (npct < 1) { if (tid == 0) *w->ncand = 0; return; }
    const float base = s_base[0], base2 = s_base[1];
    for (int i = ia + tid; i <= ib; i += 256) { s_red[i] = s_red[i] / base; s_red2[i] = s_red2[i] / base2; }
    __syncthreads();
    // --- walk the bins in descending red; each may append its +-10 peak and its +-62 peak (<= 600 ranks: serial)
    if (tid == 0) {
        int k = 0;
        for (int r = 0; r < lim; ++r) {
            const int n = s_desc[r];
            if (k >= maxcand) break;
            if (s_red[n] >= syncmin) { s_cbin[k] = n; s_clag[k] = s_jp[n]; s_csync[k] = s_red[n]; ++k; }
            if (s_jp2[n] == s_jp[n]) continue;
            if (k >= maxcand) break;
            if (s_red2[n] >= syncmin) { s_cbin[k] = n; s_clag[k] = s_jp2[n]; s_csync[k] = s_red2[n]; ++k; }
        }
        s_n = k;
    }
    __syncthreads();
    const int ncand = s_n;
    for (int i = tid; i < ncand; i += 256) { s_cf[i] = (float)s_cbin[i] * df; s_ct[i] = ((float)s_clag[i] - 0.5f) * tstep; }
    __syncthreads();
    // --- near-duplicate suppression: sequential in i (as upstream), parallel in j.  j* = first earlier candidate
    // that beats i; everything before j* that i beats is zeroed, then i itself; nothing after j* can change.
    for (int i = 1; i < ncand; ++i) {
        if (tid == 0) s_cnt[0] = 0x7fffffff;
        __syncthreads();
        const float fi = fabsf(s_cf[i]), ti = s_ct[i], si = s_csync[i];
        for (int j = tid; j < i; j += 256) {
            const float fdiff = fi - fabsf(s_cf[j]);
            const float tdiff = fabsf(ti - s_ct[j]);
            if (fabsf(fdiff) < 4.0f && tdiff < 0.04f && si < s_csync[j]) atomicMin(&s_cnt[0], j);
        }
        __syncthreads();
        const int jstar = s_cnt[0];
        for (int j = tid; j < i; j += 256) {
            if (j < jstar) {
                const float fdiff = fi - fabsf(s_cf[j]);
                const float tdiff = fabsf(ti - s_ct[j]);
                if (fabsf(fdiff) < 4.0f && tdiff < 0.04f && si >= s_csync[j]) s_csync[j] = 0.0f;
            }
        }
        if (tid == 0 && jstar != 0x7fffffff) s_csync[i] = 0.0f;
        __syncthreads();
    }
    // --- final order by rank counting: descending sync, ties ascending bin, then lag
    int nout_local = 0;
    for (int i = tid; i < ncand; i += 256) {
        const float si = s_csync[i];
        if (!(si >= syncmin)) continue;
        int rank = 0;
        for (int j = 0; j < ncand; ++j) {
            const float sj = s_csync[j];
            if (!(sj >= syncmin) || j == i) continue;
            if (sj > si || (sj == si && (s_cbin[j] < s_cbin[i] || (s_cbin[j] == s_cbin[i] && s_clag[j] < s_clag[i])))) ++rank;
        }
        if (rank < maxcand) {
            SyncChannelBuffers::Cand c;
            c.freq_bin = s_cbin[i]; c.time_step = s_clag[i]; c.sync = si; c.freq_hz = s_cf[i]; c.dt_s = s_ct[i];
            w->cand[rank] = c;
        }
        ++nout_local;
    }
    if (tid == 0) s_n = 0;
    __syncthreads();
    if (nout_local) atomicAdd(&s_n, nout_local);
    __syncthreads();
    if (tid == 0) *w->ncand = min(s_n, maxcand);
}

} // namespace cwslg

// Micro-benchmark (gfx950), round 5 step A: can the exact mode's un-fused products fl(y*h) (SSBD.hpp:167-168) run on the matrix cores?
//   (i)  BITS: v_mfma_f32_32x32x1_2b_f32 (K = 1, two 32x32 blocks) with C = inline 0 against v_mul_f32 on >= 2^32 operand pairs (random bit
//        patterns: NaN, inf, subnormals, +-0 included; and "signal-like" operands), with the only legal difference counted separately:
//        fmaf(a, b, +0) gives +0 where the product is -0.  Also confirms the operand / result lane maps the kernel design relies on.
//   (ii) RATE: cycles per tile (16 MFMAs + the block's VALU work; scripts/micro/gen_mfma_k1_loop.py) from one and two waves per SIMD, and
//        the shader clock under that load (delta s_memtime / delta s_memrealtime).
// Build: python3 gen_mfma_k1_loop.py mfma_k1_loop.inc && hipcc --offload-arch=gfx950 -O3 mfma_k1.hip -o mfma_k1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <algorithm>
#include <chrono>
#include <string>
#include "mfma_k1_loop.inc"

typedef float v32f __attribute__((ext_vector_type(32)));

__device__ __forceinline__ unsigned rng(unsigned long long &s)
{
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return (unsigned)(s >> 16);
}

// kind 0: any bit pattern; 1: finite "signal x tap" magnitudes; 2: products near the subnormal range; 3: many exact zeros and signs
__device__ __forceinline__ float draw(unsigned long long &s, int kind, int which)
{
    const unsigned r = rng(s);
    if (kind == 0) return __uint_as_float(r);
    if (kind == 1) {
        const unsigned e = which ? 100u + (rng(s) % 28u) : 120u + (rng(s) % 24u);        // taps 2^-27..2^0, samples 2^-7..2^16
        return __uint_as_float((r & 0x807fffffu) | (e << 23));
    }
    if (kind == 2) {
        const unsigned e = which ? (rng(s) % 70u) : 40u + (rng(s) % 40u);                // products around 2^-149..2^-126, operands incl. subnormals
        return __uint_as_float((r & 0x807fffffu) | (e << 23));
    }
    const unsigned t = rng(s) & 7u;
    if (t == 0) return __uint_as_float(r & 0x80000000u);                                // +-0
    if (t == 1) return __uint_as_float((r & 0x80000000u) | 0x7f800000u);                // +-inf
    return __uint_as_float((r & 0x807fffffu) | ((118u + (r >> 28)) << 23));
}

// counts[0] pairs checked, [1] bit mismatches (NaN vs NaN counted equal), [2] of the matches: -0 product delivered as +0 (expected, legal),
// [3] -0 product delivered as -0, [4] NaN results whose payload / sign differ, [5] subnormal or zero results among non-trivial products
__global__ __launch_bounds__(256) void bits_kernel(unsigned long long *counts, int iters, unsigned long long seed0)
{
    const int lane = threadIdx.x & 63;
    unsigned long long s = seed0 ^ (0x9E3779B97F4A7C15ull * (blockIdx.x * 256ull + threadIdx.x + 1));
    for (int k = 0; k < 8; ++k) rng(s);
    unsigned long long c_pairs = 0, c_bad = 0, c_z_ok = 0, c_z_neg = 0, c_nan = 0, c_sub = 0;
    for (int it = 0; it < iters; ++it) {
        const int kind = it & 3;
        const float a = draw(s, kind, 1), b = draw(s, kind, 0);
        v32f z;
#pragma unroll
        for (int r = 0; r < 32; ++r) z[r] = 0.0f;
        const v32f d = __builtin_amdgcn_mfma_f32_32x32x1f32(a, b, z, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int blk = r >> 4, rr = r & 15;
            const int row = (rr & 3) + 8 * (rr >> 2) + 4 * (lane >> 5), col = lane & 31;
            const float ai = __shfl(a, 32 * blk + row, 64), bj = __shfl(b, 32 * blk + col, 64);
            float p;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(ai), "v"(bj));
            const unsigned up = __float_as_uint(p), ud = __float_as_uint(d[r]);
            ++c_pairs;
            const bool nan_p = (up & 0x7fffffffu) > 0x7f800000u, nan_d = (ud & 0x7fffffffu) > 0x7f800000u;
            if (nan_p || nan_d) {
                if (nan_p != nan_d) ++c_bad;
                else if (up != ud) ++c_nan;
            } else if (up == 0x80000000u) {
                if (ud == 0u) ++c_z_ok;
                else if (ud == 0x80000000u) ++c_z_neg;
                else ++c_bad;
            } else if (up != ud) ++c_bad;
            if ((up & 0x7f800000u) == 0 && (__float_as_uint(ai) & 0x7fffffffu) != 0 && (__float_as_uint(bj) & 0x7fffffffu) != 0) ++c_sub;
        }
    }
    atomicAdd(&counts[0], c_pairs); atomicAdd(&counts[1], c_bad); atomicAdd(&counts[2], c_z_ok);
    atomicAdd(&counts[3], c_z_neg); atomicAdd(&counts[4], c_nan); atomicAdd(&counts[5], c_sub);
}

// ----------------------------------------------------------------------------------------------------------------------------------
#define K1_KERNEL(V)                                                                                                                 \
    __global__ __launch_bounds__(512) void rate_##V(unsigned long long *out, int iters, float seed)                                   \
    {                                                                                                                                \
        __shared__ float s_in[8 * 64 * 36];                                                                                           \
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;                                                                     \
        for (int k = threadIdx.x; k < 8 * 64 * 36; k += blockDim.x) s_in[k] = (float)((int)((k * 2654435761u) >> 17) - 16384) * seed; \
        __syncthreads();                                                                                                             \
        const unsigned la = (unsigned)(size_t)(s_in + wv * 64 * 36 + (lane & 31) * 36);                                               \
        const float x = 0.001f * (float)(lane + 1) * seed, one = 0.6f * seed, two = 0.8f * seed;                                      \
        unsigned long long t0, t1, r0, r1;                                                                                            \
        unsigned hw, xcc;                                                                                                            \
        int n = iters;                                                                                                               \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));           \
        asm volatile(                                                                                                                \
            ".irp r,0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,30,31\n\t"                       \
            "v_mov_b32 v\\r, %[x]\n\t.endr\n\t"                                                                                      \
            ".irp r,32,33,34,35,36,37,38,39,40,41,42,43,44,45,46,47,48,49,50,51,52,53,54,55,56,57,58,59,60,61,62,63\n\t"             \
            "v_mov_b32 v\\r, %[x]\n\t.endr\n\t"                                                                                      \
            ".irp r,64,65,66,67,68,69,70,71,72,73,74,75,76,77,78,79,80,81,82,83,84,85,86,87,88,89,90,91,92,93,94,95\n\t"             \
            "v_mov_b32 v\\r, %[x]\n\t.endr\n\t"                                                                                      \
            ".irp r,96,97,98,99,100,101,102,103,104,105,106,107,108,109,110,111,112,113,114,115,116,117,118,119,120,121,122,123,124,125,126,127\n\t" \
            "v_mov_b32 v\\r, %[x]\n\t.endr\n\t"                                                                                      \
            ".irp r,128,129,130,131,132,133,134,135,136,137,138,139,140,141,142,143,144,145,146,147,148,149,150,151,152,153,154,155,156,157,158,159\n\t" \
            "v_mov_b32 v\\r, %[x]\n\t.endr\n\t"                                                                                      \
            ".irp r,160,161,162,163,164,165,166,167,168,169,170,171,172,173,174,175,176,177,178,179,180,181,182,183,184,185,186,187,188,189,190,191\n\t" \
            "v_mov_b32 v\\r, %[one]\n\t.endr\n\t"                                                                                    \
            ".irp r,192,193,194,195,196,197,198,199,200,201,202,203,204,205,206,207,208,217,218,219,220,221,222,223\n\t"             \
            "v_mov_b32 v\\r, 0\n\t.endr\n\t"                                                                                         \
            "v_mov_b32 v209, %[one]\n\tv_mov_b32 v210, %[two]\n\tv_mov_b32 v211, %[one]\n\tv_mov_b32 v212, %[two]\n\t"               \
            "v_mov_b32 v213, %[one]\n\tv_mov_b32 v214, %[two]\n\tv_mov_b32 v215, %[two]\n\tv_mov_b32 v216, %[one]\n\t"               \
            "ds_read_b128 v[128:131], %[la] offset:0\n\tds_read_b128 v[132:135], %[la] offset:16\n\t"                                \
            "ds_read_b128 v[136:139], %[la] offset:32\n\tds_read_b128 v[140:143], %[la] offset:48\n\t"                               \
            "ds_read_b128 v[144:147], %[la] offset:64\n\tds_read_b128 v[148:151], %[la] offset:80\n\t"                               \
            "ds_read_b128 v[152:155], %[la] offset:96\n\tds_read_b128 v[156:159], %[la] offset:112\n\t"                              \
            "s_waitcnt lgkmcnt(0)\n\ts_barrier\n\t"                                                                                  \
            "s_memtime %[t0]\n\ts_memrealtime %[r0]\n\ts_waitcnt lgkmcnt(0)\n\t"                                                     \
            "L_loop_%=:\n\t" K1_TILE_##V                                                                                             \
            "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 L_loop_%=\n\t"                                        \
            "s_nop 15\n\ts_nop 7\n\t"                                                                                                \
            "s_memtime %[t1]\n\ts_memrealtime %[r1]\n\ts_waitcnt lgkmcnt(0)\n\t"                                                     \
            : [t0] "=&s"(t0), [t1] "=&s"(t1), [r0] "=&s"(r0), [r1] "=&s"(r1), [n] "+s"(n)                                            \
            : [x] "v"(x), [one] "v"(one), [two] "v"(two), [la] "v"(la)                                                               \
            : "memory", "scc", K1_CLOBBERS);                                                                                         \
        if (lane == 0) {                                                                                                             \
            const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + wv;                                                            \
            out[4 * w] = t0; out[4 * w + 1] = t1; out[4 * w + 2] = ((unsigned long long)xcc << 32) | hw; out[4 * w + 3] = r1 - r0;    \
        }                                                                                                                            \
    }
K1_VARIANTS(K1_KERNEL)

typedef void (*rate_fn)(unsigned long long *, int, float);

static double g_ns_tile, g_mhz;
static void run_rate(const char *name, rate_fn fn, int n_inst, int threads, int blocks, int iters, bool quiet = false)
{
    const size_t waves = (size_t)blocks * threads / 64;
    unsigned long long *d; hipMalloc(&d, waves * 32);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0f);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); exit(1); }
    std::vector<unsigned long long> h(waves * 4); hipMemcpy(h.data(), d, waves * 32, hipMemcpyDeviceToHost);
    struct Acc { unsigned long long t0 = ~0ull, t1 = 0; int n = 0; double sum = 0, clk = 0; };
    std::map<unsigned long long, Acc> by_simd;
    for (size_t w = 0; w < waves; ++w) {
        const unsigned long long id = h[4 * w + 2];
        const unsigned lo = (unsigned)id;
        const unsigned long long key = (id >> 32 << 32) | (lo & 0xFF30u) | ((lo >> 4) & 3u);
        Acc &a = by_simd[key];
        a.t0 = std::min(a.t0, h[4 * w]); a.t1 = std::max(a.t1, h[4 * w + 1]); a.n++;
        a.sum += (double)(h[4 * w + 1] - h[4 * w]);
        a.clk += (double)(h[4 * w + 1] - h[4 * w]) / (double)h[4 * w + 3] * 100.0;          // s_memrealtime ticks at 100 MHz
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    std::map<int, std::vector<double>> per_wave, per_simd, clk;
    for (auto &kv : by_simd) {
        const Acc &a = kv.second;
        per_wave[a.n].push_back(a.sum / a.n / iters);
        per_simd[a.n].push_back((double)(a.t1 - a.t0) / (a.n * (double)iters));
        clk[a.n].push_back(a.clk / a.n);
    }
    {
        std::vector<double> ns, mhz;
        for (size_t w = 0; w < waves; ++w) { ns.push_back((double)h[4 * w + 3] * 10.0 / iters); mhz.push_back((double)(h[4 * w + 1] - h[4 * w]) / (double)h[4 * w + 3] * 100.0); }
        g_ns_tile = med(ns); g_mhz = med(mhz);
    }
    if (quiet) { hipFree(d); return; }
    printf("%-10s %4d inst/tile %4d thr x %4d blk:", name, n_inst, threads, blocks);
    for (auto &kv : per_wave)
        printf("  [%d waves/SIMD on %zu SIMDs: %.0f cyc/tile per wave, %.0f cyc/tile per SIMD, %.0f MHz]", kv.first, kv.second.size(), med(kv.second),
               med(per_simd[kv.first]), med(clk[kv.first]));
    printf("\n");
    hipFree(d);
}

int main(int argc, char **argv)
{
    if (argc > 3 && std::string(argv[1]) == "sustain") {       // sustain VARIANT SECONDS: one form looping under the power controller
        const std::string v = argv[2];
        const double secs = atof(argv[3]);
        rate_fn fn = nullptr;
#define K1_PICK(V) if (v == #V) fn = rate_##V;
        K1_VARIANTS(K1_PICK)
        if (!fn) { printf("unknown variant\n"); return 2; }
        const auto t_start = std::chrono::steady_clock::now();
        int k = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() < secs) {
            run_rate(argv[2], fn, 0, 512, 256, 50000, true);
            if (++k % 4 == 0) printf("%s sustained: %.1f ns per tile per wave (two waves per SIMD), shader clock %.0f MHz, t = %.1f s\n", argv[2], g_ns_tile, g_mhz,
                                     std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
            fflush(stdout);
        }
        return 0;
    }
    const int bits_iters = argc > 1 ? atoi(argv[1]) : 512;
    {
        unsigned long long *d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
        const int blocks = 1024;                     // 1024 x 4 waves x iters x 2048 pairs: 512 iterations = 4.3e9 pairs
        hipLaunchKernelGGL(bits_kernel, dim3(blocks), dim3(256), 0, 0, d, bits_iters, 0xC0FFEEull);
        if (hipDeviceSynchronize() != hipSuccess) { printf("bits: launch failed\n"); return 1; }
        unsigned long long c[8]; hipMemcpy(c, d, 64, hipMemcpyDeviceToHost);
        printf("bits: v_mfma_f32_32x32x1_2b_f32 (C = 0) vs v_mul_f32: %llu pairs, %llu MISMATCHES, -0 products delivered as +0: %llu (as -0: %llu), "
               "NaN payload differences: %llu, results in the subnormal range from non-zero operands: %llu\n", c[0] / 64 * 64, c[1], c[2], c[3], c[4], c[5]);
        hipFree(d);
    }
    const int iters = 2000;
#define K1_RUN1(V) run_rate(#V, rate_##V, K1_NINST_##V, 256, 256, iters);
#define K1_RUN2(V) run_rate(#V, rate_##V, K1_NINST_##V, 512, 256, iters);
    K1_VARIANTS(K1_RUN1)
    K1_VARIANTS(K1_RUN2)
    return 0;
}

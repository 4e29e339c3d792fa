// Micro-benchmark (gfx950): how many VALU instructions a SIMD retires per cycle when 1, 2, 3 or 4 waves share it.
// Every wave runs the same straight-line loop (8 independent chains of v_pk_add_f32 / v_pk_mul_f32 / v_add_f32 / v_pk_fma_f32), stamps
// s_memtime before and after, and records HW_ID; the host groups waves by (XCC, SE, SH, CU, SIMD) and reports, per SIMD population,
//   per-wave cycles per instruction   and   SIMD-level cycles per instruction = (max t1 - min t0) / (instructions of all its waves).
// Build: hipcc --offload-arch=gfx950 -O3 pk_issue.hip -o pk_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void k(unsigned long long *out, float seed)
{
    constexpr int CH = 8;
    v2f a[CH];
    for (int c = 0; c < CH; ++c) a[c] = v2f{seed + c, seed - c};
    const v2f b = v2f{seed * 0.5f, seed * 0.25f};
    unsigned long long t0, t1;
    const unsigned long long sp = __builtin_amdgcn_readfirstlane(__float_as_uint(seed)) * 0x100000001ull;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c].x) : "v"(b.x));
                else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(b));
                else if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[c].x) : "v"(b.x));
                else if (KIND == 5) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[c].x) : "v"(b.x), "v"(b.y));
                else if (KIND == 6) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[c].x) : "v"(b.x));
                else if (KIND == 7) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c].x) : "v"(b.x), "v"(b.y));
                else if (KIND == 8) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(a[c].x) : "v"(b.x), "v"(b.y), "v"(a[(c + 1) % CH].y));
                else if (KIND == 9) asm volatile("v_fmac_f32_e32 %0, %2, %3\n\tv_add_f32_e32 %1, %2, %1" : "+v"(a[c].x), "+v"(a[c].y) : "v"(b.x), "v"(b.y));
                else if (KIND == 10) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(a[c]) : "v"(b), "v"(a[(c + 3) % CH]));
                else if (KIND == 11) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[c].x) : "s"(seed), "v"(b.y));
                else if (KIND == 12) asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(a[c].x) : "v"(b.x));
                else if (KIND == 13) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(a[c].x) : "s"(seed));
                else if (KIND == 14) asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(a[c].x) : "s"(seed));
                else if (KIND == 15) asm volatile("v_mul_f32_e32 %0, 0x3b5a740e, %0" : "+v"(a[c].x));
                else if (KIND == 16) asm volatile("v_mul_f32_e32 %0, 2.0, %0" : "+v"(a[c].x));
                else if (KIND == 17) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[c].x) : "s"(seed), "v"(b.y));
                else if (KIND == 18) asm volatile("v_mul_f32_e64 %0, %1, -%0" : "+v"(a[c].x) : "v"(b.x));
                else if (KIND == 19) asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(a[c].x) : "v"(b.x));
                else if (KIND == 20) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[c]) : "s"(sp), "v"(b));
                else asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[c]) : "s"(sp));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int c = 0; c < CH; ++c) s += a[c].x + a[c].y;
    if ((threadIdx.x & 63) == 0) {
        const size_t w = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        out[4 * w] = t0; out[4 * w + 1] = t1; out[4 * w + 2] = ((unsigned long long)xcc << 32) | hw; out[4 * w + 3] = (unsigned long long)s;
    }
}

template <int KIND>
void run(const char *name, int threads, int blocks)
{
    const size_t waves = (size_t)blocks * threads / 64;
    unsigned long long *d; hipMalloc(&d, waves * 32);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(waves * 4); hipMemcpy(h.data(), d, waves * 32, hipMemcpyDeviceToHost);
    const double n_inst = 256.0 * 8 * 8 * (KIND == 9 ? 2 : 1);
    struct Acc { unsigned long long t0 = ~0ull, t1 = 0; int n = 0; double sum = 0; };
    std::map<unsigned long long, Acc> by_simd;
    for (size_t w = 0; w < waves; ++w) {
        const unsigned long long id = h[4 * w + 2];
        const unsigned lo = (unsigned)id;
        const unsigned long long key = (id >> 32 << 32) | (lo & 0xFF30u) | ((lo >> 4) & 3u);   // xcc | se/sh/cu bits [15:8] | simd [5:4]
        Acc &a = by_simd[key];
        a.t0 = std::min(a.t0, h[4 * w]); a.t1 = std::max(a.t1, h[4 * w + 1]); a.n++; a.sum += (double)(h[4 * w + 1] - h[4 * w]);
    }
    std::map<int, std::vector<double>> per_wave, per_simd;
    for (auto &kv : by_simd) {
        const Acc &a = kv.second;
        per_wave[a.n].push_back(a.sum / a.n / n_inst);
        per_simd[a.n].push_back((double)(a.t1 - a.t0) / (a.n * n_inst));
    }
    printf("%-10s %4d threads x %4d blocks:", name, threads, blocks);
    for (auto &kv : per_wave) {
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("  [%d waves/SIMD on %zu SIMDs: %.2f cyc/inst per wave, %.2f cyc/inst per SIMD]", kv.first, kv.second.size(), med(kv.second), med(per_simd[kv.first]));
    }
    printf("\n");
    hipFree(d);
}
int main()
{
    for (int t : {512, 1024}) { run<0>("pk_add", t, 256); run<1>("pk_mul", t, 256); run<2>("add", t, 256); run<3>("pk_fma", t, 256); run<4>("fma", t, 256);
        run<5>("fmac_e32", t, 256); run<6>("mul_e32", t, 256); run<7>("fma_acc", t, 256); run<8>("fma_3src", t, 256); run<9>("fmac+add(x2)", t, 256);
        run<10>("pk_fma_mods", t, 256); run<11>("fmac_sgpr", t, 256); run<12>("sub_e32", t, 256);
        run<13>("mul_sgpr", t, 256); run<14>("add_sgpr", t, 256); run<15>("mul_literal", t, 256); run<16>("mul_inline2.0", t, 256); run<17>("fma_sgpr", t, 256);
        run<18>("mul_e64_neg", t, 256); run<19>("fmac_same", t, 256); run<20>("pk_fma_sgpr", t, 256); run<21>("pk_mul_sgpr", t, 256); }
    return 0;
}

// Micro-benchmark (gfx950): issue cost and dependent latency of v_pk_add_f32 / v_pk_mul_f32 / v_add_f32 from ONE wave per SIMD and from
// two: C independent chains of N dependent instructions each, timed with s_memtime.  Build: hipcc --offload-arch=gfx950 -O3 pk_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int CHAINS, int KIND>
__global__ void k(unsigned long long *out, float seed)
{
    v2f a[CHAINS];
    for (int c = 0; c < CHAINS; ++c) a[c] = v2f{seed + c, seed - c};
    const v2f b = v2f{seed * 0.5f, seed * 0.25f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                else if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[c].x) : "v"(b.x));
                else { v2f p; asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_add_f32 %3, %3, %0" : "=&v"(p) : "v"(a[c]), "v"(b), "v"(a[c])); }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += a[c].x + a[c].y;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = (unsigned long long)s; }
}

template <int CHAINS, int KIND>
void run(const char *name, int threads)
{
    unsigned long long *d; hipMalloc(&d, 16 * 1024);
    hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipLaunchKernelGGL((k<CHAINS, KIND>), dim3(256), dim3(threads), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(512); hipMemcpy(h.data(), d, 4096, hipMemcpyDeviceToHost);
    double t = 0; for (int b = 0; b < 256; ++b) t += (double)h[2 * b];
    t /= 256;
    printf("%-14s chains %d, %d waves/SIMD: %.2f ticks per instruction (%.2f per chain step)\n", name, CHAINS, threads / 256, t / (64.0 * 16 * CHAINS),
           t / (64.0 * 16));
    hipFree(d);
}
int main()
{
    // 256 threads = 4 waves = one per SIMD; 512 = two per SIMD
    run<1, 0>("pk_add", 256); run<2, 0>("pk_add", 256); run<4, 0>("pk_add", 256); run<8, 0>("pk_add", 256);
    run<1, 1>("pk_mul", 256); run<4, 1>("pk_mul", 256);
    run<1, 2>("add", 256); run<2, 2>("add", 256); run<4, 2>("add", 256); run<8, 2>("add", 256);
    run<1, 0>("pk_add", 512); run<2, 0>("pk_add", 512); run<4, 0>("pk_add", 512); run<8, 0>("pk_add", 512);
    run<4, 2>("add", 512); run<8, 2>("add", 512);
    run<8, 0>("pk_add", 1024); run<8, 2>("add", 1024);
    return 0;
}

/* Test helper (compiled by tests/test_div6_shortcut.py): compares the three-operation x / 6 of sync_kernels.hpp (div6_exact) with the
 * IEEE division for every float whose bit pattern is start + k * stride.  Prints the number of mismatches inside the range the kernel
 * uses the shortcut for (2^-95 <= |x| <= FLT_MAX) and the number outside it. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1, start = argc > 2 ? strtoull(argv[2], 0, 10) : 0;
    const float r = 0x1.555556p-3f;
    uint64_t bad_in = 0, bad_out = 0, n = 0;
    for (uint64_t u = start; u < (1ull << 32); u += stride, ++n) {
        const uint32_t b = (uint32_t)u;
        float x;
        memcpy(&x, &b, 4);
        const float ref = x / 6.0f, q0 = x * r, e = fmaf(-6.0f, q0, x), q = fmaf(e, r, q0);
        uint32_t a, c;
        memcpy(&a, &ref, 4);
        memcpy(&c, &q, 4);
        if (a == c || (ref != ref && q != q)) continue;
        const uint32_t ex = b & 0x7f800000u;
        const int shortcut = !((uint32_t)((b & 0x7fffffffu) - 0x10000000u) >= 0x6f800000u);
        if (shortcut) ++bad_in; else ++bad_out;
    }
    printf("%llu %llu %llu\n", (unsigned long long)n, (unsigned long long)bad_in, (unsigned long long)bad_out);
    return bad_in != 0;
}

// lab/demod_lab_kernels.hpp -- demod kernels that were measured and retired (rounds 1-3).  Included by cwsl_gpu.hip only with -DCWSLG_LAB=1
// (libcwslgpu_lab.so, CWSLG_DEMOD_VARIANT); the product library does not contain them and its translation unit compiles without this directory.
// They use the helpers of demod_kernels.hpp (geometry, tile context, load issue, DPP reductions), which must be included first.
#pragma once
namespace cwslg {

// ---------------------------------------------------------------------------------------------
// demod_mfma1p_kernel (D = 16; measured alternative, CWSLG_DEMOD_VARIANT=4..7 = workgroups per CU the registers are capped
// for): the tile kernel with its FIR on the matrix cores and ONE plane resident at a time.
// Per plane the polyphase FIR  out[p] = sum_{u<32} sum_{w<16} X[u][p+w] H[u][w]  splits into a dense product over the 32
// branches,  Dm[q][w] = sum_u X[u][q] H[u][w]   (143 columns q x 16 taps w, K = 32: v_mfma_f32_16x16x4_f32, A = 16 columns of
// four branches straight out of the pair-row LDS image, B = four rows of the 32 x 16 tap matrix held in 8 registers), and a
// diagonal sum  out[p] = sum_w Dm[p+w][w]  (16 conflict-free LDS reads per output).  89 % of the MFMA work is useful (143 x 16 x
// 32 against 128 x 512 products per plane); the 544 FIR v_fma per wave, their DPP reduction and the 16-tap register file
// leave the VALU (474 instead of ~950 VALU instructions per wave).  The tone-mixed samples stay in the load registers: the
// Re plane is written and multiplied, then the Im plane reuses the same LDS -- 21 KB instead of 39.7 KB per workgroup, six
// workgroups per CU instead of four.  Accumulation order differs from demod_kernel's (branches first, then taps): same
// tolerance class (4.6e-7 of frame peak against the reference-order arithmetic).
// Result: 2.71 ms against demod_kernel's 2.72 ms on the same box -- half the VALU work and 1.5x the occupancy buy nothing.
// With the MFMAs compiled out (mix, LDS traffic, barriers, diagonal sums all still there) the kernel runs at the memory
// floor of ring_probe_kernel (2.10 ms); the MFMAs add back 0.7 ms = their own pipe time (144 x 32 cycles per tile): the FIR's
// execution time, on either pipe, is what sits on top of the HBM floor.
template <int T, int NT, int WGS>
__global__ __launch_bounds__(NT, WGS) void demod_mfma1p_kernel(const ChanWork *__restrict__ works,
                                                               const float *__restrict__ taps,
                                                               int tiles_x, int n_ch)
{
    constexpr int D = 16;
    using Geo = DemodGeom<D, T>;
    constexpr int G = Geo::G;
    constexpr int PR = Geo::PR;
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    constexpr int NRB = (T / 2 + 15 + 15) / 16;          // 9 row blocks of 16 columns per plane
    constexpr int PPW = (NRB + NT / 64 - 1) / (NT / 64);     // row blocks per wave and plane: 3 (wave 0) / 2
    constexpr int PW = T / 2 + 17;                       // pitch of the product image [w][q]
    static_assert(T == 256 && NT == 256 && 2 * 16 * PW <= Geo::PLANE_FLOATS, "geometry");

    __shared__ __attribute__((aligned(16))) float s_plane[Geo::PLANE_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_aux[Geo::AUX_FLOATS];
    float2 *s_phase = reinterpret_cast<float2 *>(s_aux);

    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= min(((int)(blockIdx.x & 7) + 1) * per_xcd, total)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    TileCtx<D, T> cur;
    int ich, itile;
    item_to_ch_tile(item, tiles_x, n_ch, ich, itile);
    decode_item<D, T>(works + ich, itile, cur);
    v4f xs[NIT];
    float2 ck;
    v4f tn;
    issue_tile_loads<D, T, NT>(cur, tid, xs, ck, tn);
    float hb[8];                                          // B operand: H[4 ks + (lane >> 4)][lane & 15]
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) hb[ks] = taps[G * (lane & 15) + 4 * ks + (lane >> 4)];

    const int r0 = 2 * tid;
    float *p0 = s_plane + ((r0 % G) >> 1) * PR + 2 * (r0 / G);
    const int rel1 = r0 - D + 2 * NT;
    float *p1 = s_plane + ((rel1 % G) >> 1) * PR + 2 * (rel1 / G - (2 * NT) / G);      // the same storage, plane-1 addressing
    constexpr int WSTEP = (2 * NT) / G;

    {   // ---- phase 0: bit-exact phasor for the tile's T+31 blocks
        const int cidx = cur.ck_first + tid;
        if (tid < Geo::NCK && cidx >= 0) {
            float2 p = ck;
            const int pbase = cur.pb0 + kCk * tid;
#pragma unroll
            for (int s = 0; s < kCk; ++s) {
                const int pb = pbase + s;
                if (pb >= 0 && pb < Geo::NBLK) s_phase[pb] = p;
                p = cmul_exact(p, cur.inc);
            }
        }
    }
    lds_barrier();
    if (cur.n_out == 0) return;
    const int fv = cur.first_valid;
    // A-operand / product addresses of this wave's row blocks (the same for both planes)
    const float *pa[PPW];
    float *pd[PPW];
    bool on[PPW];
    {
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const int q = lane & 15, kk = lane >> 4;
        const int la = (kk >> 1) * PR + (kk & 1) + 2 * q;
        const int ld = (lane & 15) * PW + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int rb0 = wvu + (NT / 64) * i;
            on[i] = rb0 < NRB;
            const int rb = on[i] ? rb0 : 0;                 // a wave without a third block repeats block 0 and drops it
            pa[i] = s_plane + 32 * rb + la;
            pd[i] = s_plane + 16 * rb + ld;
        }
    }
    // ---- phase 1a: tone mix (kept in the load registers), Re part -> the plane
    {
        const float2 tn0 = make_float2(tn.x, tn.y), tn1 = make_float2(tn.z, tn.w);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = 2 * tid + it * 2 * NT;
            int blk = (2 * tid) / D + it * (2 * NT / D);
            if (blk > Geo::NBLK - 1) blk = Geo::NBLK - 1;
            const float2 ph = s_phase[blk];
            const v4f x = xs[it];
            v4f m;
            m.x = __builtin_fmaf(x.x, tn0.x, -(x.y * tn0.y));
            m.y = __builtin_fmaf(x.x, tn0.y, x.y * tn0.x);
            m.z = __builtin_fmaf(x.z, tn1.x, -(x.w * tn1.y));
            m.w = __builtin_fmaf(x.z, tn1.y, x.w * tn1.x);
            xs[it] = m;
            float y0r = __builtin_fmaf(m.x, ph.x, -(m.y * ph.y));
            float y1r = __builtin_fmaf(m.z, ph.x, -(m.w * ph.y));
            if (fv != 0) { if (r < fv) { y0r = 0.f; y1r = 0.f; } }
            const bool in0 = (it < NIT - 1) || (r < G * (T / 2 + 15));
            if (in0) *reinterpret_cast<float2 *>(p0 + 2 * it * WSTEP) = make_float2(y0r, y1r);
        }
    }
    lds_barrier();
    v4f acc0[PPW], acc1[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) { acc0[i] = v4f{0.f, 0.f, 0.f, 0.f}; acc1[i] = v4f{0.f, 0.f, 0.f, 0.f}; }
    {   // all A operands first (24 LDS reads in flight), then the MFMA stream: a read issued right before its MFMA
        // exposes the LDS latency 24 times per plane
        float av[8][PPW];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int i = 0; i < PPW; ++i) av[ks][i] = pa[i][2 * PR * ks];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int i = 0; i < PPW; ++i) acc0[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks][i], hb[ks], acc0[i], 0, 0, 0);
    }
    lds_barrier();                                        // the Re plane has been consumed
    // ---- phase 1b: Im part -> the same storage (plane-1 alignment: relative sample r - D)
    {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = 2 * tid + it * 2 * NT;
            int blk = (2 * tid) / D + it * (2 * NT / D);
            if (blk > Geo::NBLK - 1) blk = Geo::NBLK - 1;
            const float2 ph = s_phase[blk];
            const v4f m = xs[it];
            float y0i = __builtin_fmaf(m.x, ph.y, m.y * ph.x);
            float y1i = __builtin_fmaf(m.z, ph.y, m.w * ph.x);
            if (fv != 0) { if (r < fv) { y0i = 0.f; y1i = 0.f; } }
            const bool in1 = (it > 0) ? ((it < NIT - 1) || (r < Geo::NSAMP)) : (r >= D);
            if (in1) *reinterpret_cast<float2 *>(p1 + 2 * it * WSTEP) = make_float2(y0i, y1i);
        }
    }
    lds_barrier();
    {
        float av[8][PPW];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int i = 0; i < PPW; ++i) av[ks][i] = pa[i][2 * PR * ks];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int i = 0; i < PPW; ++i) acc1[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks][i], hb[ks], acc1[i], 0, 0, 0);
    }
    lds_barrier();                                        // the Im plane has been consumed; s_phase is no longer needed either
    // ---- products -> LDS as [plane][w][q] over the plane storage, then the diagonal sums
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        if (on[i]) {
            pd[i][0] = acc0[i].x; pd[i][1] = acc0[i].y; pd[i][2] = acc0[i].z; pd[i][3] = acc0[i].w;
            float *d1 = pd[i] + 16 * PW;
            d1[0] = acc1[i].x; d1[1] = acc1[i].y; d1[2] = acc1[i].z; d1[3] = acc1[i].w;
        }
    lds_barrier();
    {
        const int pl = tid >> 7, p = tid & 127;
        const float *src = s_plane + pl * 16 * PW + p;
        float sum = 0.0f;
#pragma unroll
        for (int wq = 0; wq < 16; ++wq) sum = sum + src[wq * (PW + 1)];
        const float sgn_plane = pl ? -cur.sign : 1.0f;
        const float sg = (p & 1) ? -sgn_plane : sgn_plane;
        s_aux[2 * p + pl] = sg * sum;
    }
    lds_barrier();
    {
        CWSLG_GLOBAL float *out = as_global_rw(cur.out) + (size_t)cur.tile * T;
        float mx = 0.0f;
        for (int o = tid; o < cur.n_out; o += NT) {
            const float v = s_aux[o];
            out[o] = v;
            mx = fmaxf(mx, fabsf(v));
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) publish_peak(cur.peak, mx);
    }
}

// ---------------------------------------------------------------------------------------------
// demod_mfma_bf16_kernel (D = 16; CWSLG_DEMOD_VARIANT=8): demod_mfma1p_kernel's structure with the dense product on the BF16 matrix
// cores at fp32 accuracy.  demod_kernel holds the package at its 1400 W limit (profiles/r2_power.txt): what separates it from the
// floor of its access pattern is the ENERGY of the fp32 FMAs, and a bf16 MFMA costs a small fraction of an fp32 FMA per product.
// Every mixed sample y and every tap h is split into three bf16 terms,  y = a + b + c,  h = d + e + f  (a = bf16(y), b = bf16(y - a),
// c = bf16(y - a - b): the residuals are exact in fp32, so the three terms carry all 24 bits), and
//     y h  =  a d + a e + b d + a f + b e + c d   (+ b f + c e + c f, below 2^-24 of the product: dropped)
// Each bf16 x bf16 product is exact in fp32 and the MFMA accumulates in fp32, so the result has fp32-class rounding (measured
// against the reference-order arithmetic by the same tests as demod_kernel).  Per 16-column row block and plane: six
// v_mfma_f32_16x16x32_bf16 (K = 32 = the branches, one instruction each) instead of eight v_mfma_f32_16x16x4_f32 at 16 instead of
// 32 cycles: 0.375 of the matrix-pipe time.  LDS image per plane: [split][column q][branch u] bf16, 80-byte column pitch (the 16
// lanes of a ds_read_b128 group start 20 banks apart: conflict-free), 34.5 KB, one plane resident at a time: 4 workgroups per CU.
// Result (same box, 512 slots, demod only): numerics as hoped -- 4.6e-7 of frame peak, every parity test of the default mode green --
// and 2.67 ms at 1400 W / 1.97 GHz against demod_kernel's 2.58 ms at 1400 W / 1.90 GHz and the f32 matrix-core form's 2.59 ms at
// 1400 W / 1.99 GHz.  Three different ways of doing the FIR, one launch time and one power reading: the FIR's arithmetic is NOT what
// holds the package at its limit.  Kept as a measured alternative.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// (y0, y1) -> three packed bf16 pairs (round to nearest even); the residuals y - a and y - a - b are exact
__device__ __forceinline__ void split3_bf16(float y0, float y1, unsigned &a, unsigned &b, unsigned &c)
{
    auto pk = [](float u0, float u1) {
        const bf16x2 t = __builtin_convertvector(v2f{u0, u1}, bf16x2);
        return __builtin_bit_cast(unsigned, t);
    };
    a = pk(y0, y1);
    const float f0 = y0 - __uint_as_float(a << 16), f1 = y1 - __uint_as_float(a & 0xFFFF0000u);
    b = pk(f0, f1);
    const float g0 = f0 - __uint_as_float(b << 16), g1 = f1 - __uint_as_float(b & 0xFFFF0000u);
    c = pk(g0, g1);
}

template <int T, int NT, int WGS>
__global__ __launch_bounds__(NT, WGS) void demod_mfma_bf16_kernel(const ChanWork *__restrict__ works,
                                                                  const float *__restrict__ taps,
                                                                  int tiles_x, int n_ch)
{
    constexpr int D = 16;
    using Geo = DemodGeom<D, T>;
    constexpr int G = Geo::G;
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    constexpr int NRB = (T / 2 + 15 + 15) / 16;          // 9 row blocks of 16 columns per plane
    constexpr int PPW = (NRB + NT / 64 - 1) / (NT / 64);     // row blocks per wave and plane: 3 (wave 0) / 2
    constexpr int PW = T / 2 + 17;                       // pitch of the product image [w][q]
    constexpr int NQ = 16 * NRB;                         // 144 columns
    constexpr int QP = 80;                               // bytes per column of one split image
    constexpr int SPLIT_BYTES = NQ * QP;                 // 11 520
    static_assert(T == 256 && NT == 256 && G == 32 && 2 * 16 * PW * 4 <= 3 * SPLIT_BYTES, "geometry");

    __shared__ __attribute__((aligned(16))) unsigned char s_img[3 * SPLIT_BYTES];
    __shared__ __attribute__((aligned(16))) float s_aux[Geo::AUX_FLOATS];
    float2 *s_phase = reinterpret_cast<float2 *>(s_aux);
    float *s_prod = reinterpret_cast<float *>(s_img);

    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= min(((int)(blockIdx.x & 7) + 1) * per_xcd, total)) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    TileCtx<D, T> cur;
    int ich, itile;
    item_to_ch_tile(item, tiles_x, n_ch, ich, itile);
    decode_item<D, T>(works + ich, itile, cur);
    v4f xs[NIT];
    float2 ck;
    v4f tn;
    issue_tile_loads<D, T, NT>(cur, tid, xs, ck, tn);
    // B operand: H[u = 8 (lane >> 4) + j][w = lane & 15], j = 0..7, as three bf16 fragments
    bf16x8 hd, he, hf;
    {
        const float *hp = taps + G * (lane & 15) + 8 * (lane >> 4);
        unsigned d[4], e[4], f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) split3_bf16(hp[2 * j], hp[2 * j + 1], d[j], e[j], f[j]);
        struct U4 { unsigned v[4]; };
        hd = __builtin_bit_cast(bf16x8, U4{{d[0], d[1], d[2], d[3]}});
        he = __builtin_bit_cast(bf16x8, U4{{e[0], e[1], e[2], e[3]}});
        hf = __builtin_bit_cast(bf16x8, U4{{f[0], f[1], f[2], f[3]}});
    }

    // image addresses of this thread's sample pairs: sample r = 2 tid + 512 it -> column r / 32, branches (r % 32, r % 32 + 1)
    unsigned char *w0 = s_img + (tid >> 4) * QP + (tid & 15) * 4;                    // plane 0: rel = r
    const int rel1 = 2 * tid - D + 2 * NT;                                           // plane 1: rel = r - D, taken at it = 1
    unsigned char *w1 = s_img + ((rel1 >> 5) - (2 * NT) / G) * QP + (rel1 & 31) * 2;
    constexpr int WSTEP_B = ((2 * NT) / G) * QP;                                     // bytes per iteration: 16 columns

    {   // ---- phase 0: bit-exact phasor for the tile's T+31 blocks
        const int cidx = cur.ck_first + tid;
        if (tid < Geo::NCK && cidx >= 0) {
            float2 p = ck;
            const int pbase = cur.pb0 + kCk * tid;
#pragma unroll
            for (int s = 0; s < kCk; ++s) {
                const int pb = pbase + s;
                if (pb >= 0 && pb < Geo::NBLK) s_phase[pb] = p;
                p = cmul_exact(p, cur.inc);
            }
        }
    }
    lds_barrier();
    if (cur.n_out == 0) return;
    const int fv = cur.first_valid;
    // A-fragment / product addresses of this wave's row blocks (the same for both planes)
    const unsigned char *pa[PPW];
    float *pd[PPW];
    bool on[PPW];
    {
        const int wvu = __builtin_amdgcn_readfirstlane(wv);
        const int ld = (lane & 15) * PW + 4 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            const int rb0 = wvu + (NT / 64) * i;
            on[i] = rb0 < NRB;
            const int rb = on[i] ? rb0 : 0;                 // a wave without a third block repeats block 0 and drops it
            pa[i] = s_img + (16 * rb + (lane & 15)) * QP + 16 * (lane >> 4);
            pd[i] = s_prod + 16 * rb + ld;
        }
    }
    auto products = [&](v4f (&acc)[PPW]) {
        bf16x8 fa[PPW], fb[PPW], fc[PPW];
#pragma unroll
        for (int i = 0; i < PPW; ++i) {
            fa[i] = *reinterpret_cast<const bf16x8 *>(pa[i]);
            fb[i] = *reinterpret_cast<const bf16x8 *>(pa[i] + SPLIT_BYTES);
            fc[i] = *reinterpret_cast<const bf16x8 *>(pa[i] + 2 * SPLIT_BYTES);
        }
#if CWSLG_BF16_DIAG >= 1                                  // diagnostic builds (scripts/gpu_bf16_diag.sh): no matrix instructions
#pragma unroll
        for (int i = 0; i < PPW; ++i) acc[i].x += (float)fa[i][0] + (float)fb[i][1] + (float)fc[i][2];
        return;
#endif
#pragma unroll
        for (int i = 0; i < PPW; ++i) {                   // smallest terms first
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[i], hd, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], he, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], hf, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[i], hd, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], he, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], hd, acc[i], 0, 0, 0);
        }
    };
    // ---- phase 1a: tone mix (kept in the load registers), Re part -> the split image
    {
        const float2 tn0 = make_float2(tn.x, tn.y), tn1 = make_float2(tn.z, tn.w);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = 2 * tid + it * 2 * NT;
            int blk = (2 * tid) / D + it * (2 * NT / D);
            if (blk > Geo::NBLK - 1) blk = Geo::NBLK - 1;
            const float2 ph = s_phase[blk];
            const v4f x = xs[it];
            v4f m;
            m.x = __builtin_fmaf(x.x, tn0.x, -(x.y * tn0.y));
            m.y = __builtin_fmaf(x.x, tn0.y, x.y * tn0.x);
            m.z = __builtin_fmaf(x.z, tn1.x, -(x.w * tn1.y));
            m.w = __builtin_fmaf(x.z, tn1.y, x.w * tn1.x);
            xs[it] = m;
            float y0r = __builtin_fmaf(m.x, ph.x, -(m.y * ph.y));
            float y1r = __builtin_fmaf(m.z, ph.x, -(m.w * ph.y));
            if (fv != 0) { if (r < fv) { y0r = 0.f; y1r = 0.f; } }
            const bool in0 = (it < NIT - 1) || (r < G * (T / 2 + 15));
            if (in0) {
                unsigned a, b, c;
#if CWSLG_BF16_DIAG >= 2                                  // ... and no split arithmetic
                a = __float_as_uint(y0r); b = __float_as_uint(y1r); c = a ^ b;
#else
                split3_bf16(y0r, y1r, a, b, c);
#endif
                unsigned char *dst = w0 + it * WSTEP_B;
                *reinterpret_cast<unsigned *>(dst) = a;
                *reinterpret_cast<unsigned *>(dst + SPLIT_BYTES) = b;
                *reinterpret_cast<unsigned *>(dst + 2 * SPLIT_BYTES) = c;
            }
        }
    }
    lds_barrier();
    v4f acc0[PPW], acc1[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) { acc0[i] = v4f{0.f, 0.f, 0.f, 0.f}; acc1[i] = v4f{0.f, 0.f, 0.f, 0.f}; }
    products(acc0);
    lds_barrier();                                        // the Re image has been consumed
    // ---- phase 1b: Im part -> the same storage (plane-1 alignment: relative sample r - D)
    {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = 2 * tid + it * 2 * NT;
            int blk = (2 * tid) / D + it * (2 * NT / D);
            if (blk > Geo::NBLK - 1) blk = Geo::NBLK - 1;
            const float2 ph = s_phase[blk];
            const v4f m = xs[it];
            float y0i = __builtin_fmaf(m.x, ph.y, m.y * ph.x);
            float y1i = __builtin_fmaf(m.z, ph.y, m.w * ph.x);
            if (fv != 0) { if (r < fv) { y0i = 0.f; y1i = 0.f; } }
            const bool in1 = (it > 0) ? ((it < NIT - 1) || (r < Geo::NSAMP)) : (r >= D);
            if (in1) {
                unsigned a, b, c;
#if CWSLG_BF16_DIAG >= 2
                a = __float_as_uint(y0i); b = __float_as_uint(y1i); c = a ^ b;
#else
                split3_bf16(y0i, y1i, a, b, c);
#endif
                unsigned char *dst = w1 + it * WSTEP_B;          // (w1 already carries the one-iteration offset)
                *reinterpret_cast<unsigned *>(dst) = a;
                *reinterpret_cast<unsigned *>(dst + SPLIT_BYTES) = b;
                *reinterpret_cast<unsigned *>(dst + 2 * SPLIT_BYTES) = c;
            }
        }
    }
    lds_barrier();
    products(acc1);
    lds_barrier();                                        // the Im image has been consumed; s_phase is no longer needed either
    // ---- products -> LDS as [plane][w][q] over the image storage, then the diagonal sums
#pragma unroll
    for (int i = 0; i < PPW; ++i)
        if (on[i]) {
            pd[i][0] = acc0[i].x; pd[i][1] = acc0[i].y; pd[i][2] = acc0[i].z; pd[i][3] = acc0[i].w;
            float *d1 = pd[i] + 16 * PW;
            d1[0] = acc1[i].x; d1[1] = acc1[i].y; d1[2] = acc1[i].z; d1[3] = acc1[i].w;
        }
    lds_barrier();
    {
        const int pl = tid >> 7, p = tid & 127;
        const float *src = s_prod + pl * 16 * PW + p;
        float sum = 0.0f;
#pragma unroll
        for (int wq = 0; wq < 16; ++wq) sum = sum + src[wq * (PW + 1)];
        const float sgn_plane = pl ? -cur.sign : 1.0f;
        const float sg = (p & 1) ? -sgn_plane : sgn_plane;
        s_aux[2 * p + pl] = sg * sum;
    }
    lds_barrier();
    {
        CWSLG_GLOBAL float *out = as_global_rw(cur.out) + (size_t)cur.tile * T;
        float mx = 0.0f;
        for (int o = tid; o < cur.n_out; o += NT) {
            const float v = s_aux[o];
            out[o] = v;
            mx = fmaxf(mx, fabsf(v));
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        if (lane == 0) publish_peak(cur.peak, mx);
    }
}

// ---------------------------------------------------------------------------------------------
// ring_probe_kernel (diagnostic, CWSLG_DEMOD_VARIANT=9): the tile kernel's memory traffic without its arithmetic -- same
// work-item order, same descriptor decode, same loads (IQ tile, checkpoint, tone), same 1 KB output row per tile; the
// loaded values are only summed.  Its time is the floor the memory system sets for this access pattern.
template <int D, int T, int NT, int FLAVOUR>
__global__ __launch_bounds__(NT, 4) void ring_probe_kernel(const ChanWork *__restrict__ works, const float *__restrict__ taps,
                                                           int tiles_x, int n_ch)
{
    using Geo = DemodGeom<D, T>;
    // FLAVOUR 1: + the per-wave atomicMax on the channel's peak word; 2: + the tile kernel's LDS footprint (4 workgroups per CU);
    // 3, 4, 5: flavour 2 + the workgroup idles (s_sleep, no pipe used) for ~2k / 4k / 8k cycles after its loads have arrived
    __shared__ float s_pad[(FLAVOUR >= 2) ? 9900 : 1];
    if (FLAVOUR >= 2) s_pad[threadIdx.x] = 0.f;
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (item >= min(((int)(blockIdx.x & 7) + 1) * per_xcd, total)) return;
    const int tid = threadIdx.x;
    TileCtx<D, T> cur;
    int ich, itile;
    item_to_ch_tile(item, tiles_x, n_ch, ich, itile);
    decode_item<D, T>(works + ich, itile, cur);
    if (cur.n_out == 0) return;
    v4f xs[NIT];
    float2 ck;
    v4f tn;
    issue_tile_loads<D, T, NT>(cur, tid, xs, ck, tn);
    float s = ck.x + tn.x + taps[tid & 15];
#pragma unroll
    for (int it = CWSLG_DIAG_NOHALO; it < NIT; ++it) s += xs[it].x + xs[it].y + xs[it].z + xs[it].w;
    if (FLAVOUR >= 2) s += s_pad[(threadIdx.x * 7) % 9900];
    if (FLAVOUR >= 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int q = 0; q < (1 << (FLAVOUR - 3)); ++q) __builtin_amdgcn_s_sleep(32);      // 32 x 64 = 2048 cycles each
    }
    if (tid < cur.n_out) as_global_rw(cur.out)[(size_t)cur.tile * T + tid] = s;
    if (FLAVOUR == 1 && (tid & 63) == 0) publish_peak(cur.peak, fabsf(s));
}

// ---------------------------------------------------------------------------------------------
// demod_exact_kernel: the reference's arithmetic, operation for operation (SSBD.hpp:160-183), for bit-exact
// verification.  Same tiles, same HBM layout, same phasor rebuild as demod_kernel; what differs is phase 2:
//   thread = one output b.  For n = 0..31 (oldest block first, like the workspace slot's accumulation order):
//       sum  = 0;  for m = 0..D-1:  sum += (x[D(b-31+n)+m] * tone[m]) * h[m + D n]      (complex*complex, complex*real, +=)
//       ws  += sum * phase_{b-31+n}
//   audio[b] = the Re/Im pick of Iterate() (SSBD.hpp:132-135)
// every product and sum un-fused (-ffp-contract=off), so the float frame -- and therefore the int16 frame -- equals the
// compiled reference bit for bit.  x*tone is computed once per sample into LDS (pitch D+1 complex per block: the 64
// lanes' reads land in 32 distinct bank pairs).  ~2.3x the VALU work and 14x the LDS traffic of demod_kernel: this is
// the checking mode (cwslg_set_exact), not the throughput path.
template <int D, int T, int NT>
__global__ __launch_bounds__(NT) void demod_exact_kernel(const ChanWork *__restrict__ works,
                                                          const float *__restrict__ taps,
                                                          int tiles_x, int n_ch)
{
    using Geo = DemodGeom<D, T>;
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    static_assert(NT == T, "one thread per output");
    __shared__ float2 s_t[Geo::NBLK * (D + 1)];
    __shared__ float2 s_phase[Geo::NBLK + 1];

    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wid >= total) return;
    const int tid = threadIdx.x;
    TileCtx<D, T> cur;
    int ich, itile;
    item_to_ch_tile(wid, tiles_x, n_ch, ich, itile);
    decode_item<D, T>(works + ich, itile, cur);
    if (cur.n_out == 0) return;
    v4f xs[NIT];
    float2 ck;
    v4f tn;
    issue_tile_loads<D, T, NT>(cur, tid, xs, ck, tn);
    {
        const int cidx = cur.ck_first + tid;
        if (tid < Geo::NCK && cidx >= 0) {
            float2 p = ck;
            const int pbase = cur.pb0 + kCk * tid;
#pragma unroll
            for (int s = 0; s < kCk; ++s) {
                const int pb = pbase + s;
                if (pb >= 0 && pb < Geo::NBLK) s_phase[pb] = p;
                p = cmul_exact(p, cur.inc);
            }
        }
    }
    // t = in[m] * tone[m]  (SSBD.hpp:167), un-fused
    {
        const float2 tn0 = make_float2(tn.x, tn.y), tn1 = make_float2(tn.z, tn.w);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = 2 * tid + it * 2 * NT;
            if (r < Geo::NSAMP) {
                const v4f x = xs[it];
                const float2 a = cmul_exact(make_float2(x.x, x.y), tn0);
                const float2 b = cmul_exact(make_float2(x.z, x.w), tn1);
                const int blk = r / D, m = r % D;
                s_t[blk * (D + 1) + m] = a;
                s_t[blk * (D + 1) + m + 1] = b;
            }
        }
    }
    __syncthreads();
    if (tid < cur.n_out) {
        const int o = tid;                                   // output qs + o ; its blocks are tile blocks o .. o+31
        const int first_blk = cur.first_valid / D;           // tile blocks before this precede the demodulator's origin
        float wr = 0.0f, wi = 0.0f;                          // the workspace slot, zero after its last read-out (:178)
        for (int n = 0; n < 32; ++n) {
            const int blk = o + n;
            if (blk < first_blk) continue;                   // the reference never touched the slot for these
            const float2 *tp = s_t + blk * (D + 1);
            float sr = 0.0f, si = 0.0f;
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float2 t = tp[m];
                const float h = taps[m + D * n];
                sr = sr + t.x * h;
                si = si + t.y * h;
            }
            const float2 ph = s_phase[blk];
            const float2 pr = cmul_exact(make_float2(sr, si), ph);      // sum * phase (:170)
            wr = wr + pr.x;
            wi = wi + pr.y;
        }
        // Iterate(): out[k] for block index mod 4 (qs is a multiple of 4; T is too)
        float v;
        switch (o & 3) {
        case 0: v = wr; break;
        case 1: v = -wi * cur.sign; break;
        case 2: v = -wr; break;
        default: v = wi * cur.sign; break;
        }
        as_global_rw(cur.out)[(size_t)cur.tile * T + o] = v;
        float mx = fabsf(v);
#pragma unroll
        for (int msk = 32; msk >= 1; msk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, msk, 64));
        if ((tid & 63) == 0) publish_peak(cur.peak, mx);
    }
}

// ---------------------------------------------------------------------------------------------
// demod_exact2_kernel: the same arithmetic as demod_exact_kernel, operation for operation (bit-identical frames), arranged after
// its counters (LDS array 86 % busy, 16 waves per CU, 7.7 ms per 512 slots = 19.7 % of HBM peak):
//   * one thread computes TWO adjacent outputs o, o + 1.  Their 32-block windows share 31 blocks, so each block's 16 mixed
//     samples are read from LDS once and used twice (tap block n for o, n - 1 for o + 1): half the LDS reads per output;
//   * even and odd blocks live in separate LDS arrays of pitch D + 2 (a row = the block's D mixed samples, then its mixer phase in
//     the pad slot): lane l reads block 2 l + n, i.e. row l + n/2 of the array of parity n & 1, as ds_read_b128 whose 16-lane groups hit
//     16 distinct 16-byte slots;
//   * the 16 taps of a block are fetched by VECTOR loads from a lane-invariant address (L1 broadcast) one step ahead: they are
//     counted in vmcnt, so waiting for them does not drain the LDS reads in flight the way scalar loads (lgkmcnt) did in the
//     round-1 attempt at this layout.
// Tile = 248 outputs on 128 threads (124 active): 39.7 KB of LDS, so FOUR tiles = 8 waves fit a CU -- two waves per SIMD; with
// 256-output tiles (41.7 KB, 3 per CU) this kernel ran no faster than the one-output form.  A persistent form with register prefetch
// of the next tile was measured no faster.
// The un-fused order needs 150 VALU lane-operations per input sample against demod_kernel's 44, so its ceiling is ~40 % of the
// HBM roofline at full VALU rate; this is the mode whose int16 frames -- and therefore candidate lists -- equal the reference
// chain's bit for bit (tests/test_gpu_exact.py, tests/test_gpu_e2e_candidates.py).
template <int D, int T, int NT>
__global__ __launch_bounds__(NT, 2) void demod_exact2_kernel(const ChanWork *__restrict__ works,
                                                              const float *__restrict__ taps,
                                                              int tiles_x, int n_ch)
{
    using Geo = DemodGeom<D, T>;
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    constexpr int NBH = (Geo::NBLK + 1) / 2 + 1;          // blocks per parity array (+1 slack)
    constexpr int BP = D + 2;                             // block pitch in complex: even (16-byte aligned pairs -> ds_read_b128), and lane
                                                          // stride 2 (D + 2) dwords = 36 (mod 64) for D = 16: the 16 lanes of a b128 group hit 16
                                                          // distinct 16-byte slots
    static_assert(2 * NT >= T && T % 4 == 0 && D % 4 == 0, "two outputs per thread");
    // row b >> 1 of array b & 1 = block b: its D mixed samples, then (slot D) the mixer phase of the block -- the pad that makes the
    // pitch even carries the one other per-block value the FIR step reads, so there is no separate phase array and a 248-output
    // tile fits four to a CU
    __shared__ __attribute__((aligned(16))) float2 s_t[2][NBH * BP];
    static_assert(sizeof(float2) * 2 * NBH * BP <= 40960, "four tiles per CU");

    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int wid = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wid >= min(((int)(blockIdx.x & 7) + 1) * per_xcd, total)) return;
    const int tid = threadIdx.x;
    TileCtx<D, T> cur;
    int ich, itile;
    item_to_ch_tile(wid, tiles_x, n_ch, ich, itile);
    decode_item<D, T>(works + ich, itile, cur);
    if (cur.n_out == 0) return;
    v4f xs[NIT];
    float2 ck;
    v4f tn;
    issue_tile_loads<D, T, NT>(cur, tid, xs, ck, tn);
    // taps through the vector path: an address the compiler cannot prove uniform (it is: every lane adds 0)
    int lane_zero = 0;
    asm volatile("" : "+v"(lane_zero));
    const CWSLG_GLOBAL v4f *tapv = as_global(reinterpret_cast<const v4f *>(taps)) + lane_zero;
    v4f hnext[D / 4];
#pragma unroll
    for (int q = 0; q < D / 4; ++q) hnext[q] = tapv[q];   // tap block 0
    {
        for (int lt = tid; lt < Geo::NCK; lt += NT) {
            const int cidx = cur.ck_first + lt;
            if (cidx >= 0) {
                const v2f t = as_global(reinterpret_cast<const v2f *>(cur.ckpt))[cidx];
                float2 p = make_float2(t.x, t.y);
                const int pbase = cur.pb0 + kCk * lt;
#pragma unroll
                for (int s = 0; s < kCk; ++s) {
                    const int pb = pbase + s;
                    if (pb >= 0 && pb < Geo::NBLK) s_t[pb & 1][(pb >> 1) * BP + D] = p;
                    p = cmul_exact(p, cur.inc);
                }
            }
        }
    }
    // t = in[m] * tone[m]  (SSBD.hpp:167), un-fused, into the parity arrays
    {
        const float2 tn0 = make_float2(tn.x, tn.y), tn1 = make_float2(tn.z, tn.w);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = 2 * tid + it * 2 * NT;
            if (r < Geo::NSAMP) {
                const v4f x = xs[it];
                const float2 a = cmul_exact(make_float2(x.x, x.y), tn0);
                const float2 b = cmul_exact(make_float2(x.z, x.w), tn1);
                const int blk = r / D, m = r % D;
                v4f ab; ab.x = a.x; ab.y = a.y; ab.z = b.x; ab.w = b.y;          // m is even: one aligned 16-byte store
                *reinterpret_cast<v4f *>(&s_t[blk & 1][(blk >> 1) * BP + m]) = ab;
            }
        }
    }
    __syncthreads();
    const int o0 = 2 * tid;
    if (o0 < T && o0 < cur.n_out) {
        const int first_blk = cur.first_valid / D;           // tile blocks before this precede the demodulator's origin
        // Software pipeline, unrolled by six (two sample buffers x three tap buffers rotate by renaming): while step n is computed,
        // the LDS reads of block n + 1 and the tap loads of block n + 1 are in flight.  (Occupancy is LDS-bound -- three tiles, six
        // waves per CU -- so registers are plentiful and latency has to be hidden inside the wave.)
        // Everything per block is written on <2 x float> values (re, im): the products (re*h, im*h) are ONE v_pk_mul_f32, the running
        // sums ONE v_pk_add_f32 -- the same IEEE operations on the same operands, half the instructions.  Packed f32 has no higher
        // FLOP rate than scalar f32, but with two waves per SIMD (LDS-bound occupancy) the kernel is ISSUE-bound, and a packed
        // instruction keeps the pipe busy for two issue slots.
        auto load_block = [&](int n, v2f (&t)[D], v2f &ph) {
            const v4f *tp = reinterpret_cast<const v4f *>(&s_t[n & 1][(tid + (n >> 1)) * BP]);
#pragma unroll
            for (int m = 0; m < D; m += 2) {
                const v4f q = tp[m >> 1];
                t[m] = v2f{q.x, q.y};
                t[m + 1] = v2f{q.z, q.w};
            }
            ph = *reinterpret_cast<const v2f *>(&s_t[n & 1][(tid + (n >> 1)) * BP + D]);
        };
        auto load_taps = [&](int n, v4f (&h)[D / 4]) {
#pragma unroll
            for (int q = 0; q < D / 4; ++q) h[q] = tapv[(D / 4) * (n < 32 ? n : 31) + q];
        };
        auto accumulate = [&](const v2f (&t)[D], const v4f (&h)[D / 4], v2f ph, v2f &w) {
            v2f sum = {0.0f, 0.0f};
#pragma unroll
            for (int m = 0; m < D; ++m) {
                const float hm = h[m >> 2][m & 3];
                const v2f hh = {hm, hm};
                sum = sum + t[m] * hh;                       // sr += t.x*h ; si += t.y*h   (:167-168)
            }
            // sum * phase (:170), std::complex's (ac - bd, ad + bc): ac, ad from sum.x, bd, bc from sum.y
            const v2f sxx = {sum.x, sum.x}, syy = {sum.y, sum.y}, phs = {ph.y, ph.x};
            const v2f p1 = sxx * ph;                         // (ac, ad)
            const v2f p2 = syy * phs;                        // (bd, bc)
            const v2f pr = {p1.x - p2.x, p1.y + p2.y};
            w = w + pr;
        };
        // step n: block o0 + n feeds output o0 with tap block n (n <= 31) and output o0 + 1 with tap block n - 1 (n >= 1)
        v2f w0 = {0.0f, 0.0f}, w1 = {0.0f, 0.0f};           // the two workspace slots, zero after their last read-out (:178)
        auto step = [&](int n, const v2f (&t)[D], v2f ph, const v4f (&hn)[D / 4], const v4f (&hnm1)[D / 4]) {
            if (o0 + n >= first_blk) {
                if (n <= 31) accumulate(t, hn, ph, w0);
                if (n >= 1) accumulate(t, hnm1, ph, w1);
            }
        };
        // taps[n] lives in buffer n mod 3: at step n the current block is buffer n mod 3, the previous one (n - 1) mod 3, and the
        // third is free for the prefetch of taps[n + 1]; the mixed samples ping-pong between tA and tB.
        v2f tA[D], tB[D], phA, phB;
        v4f h0[D / 4], h1[D / 4], h2[D / 4];
#pragma unroll
        for (int q = 0; q < D / 4; ++q) { h0[q] = hnext[q]; h2[q] = hnext[q]; }   // tap block 0 (fetched at the top); h2 is a placeholder for "block -1"
        load_block(0, tA, phA);
#pragma unroll 1
        for (int n = 0; n < 33; n += 6) {                      // 33 steps = 5 x 6 + 3
            load_taps(n + 1, h1); load_block(n + 1, tB, phB); step(n, tA, phA, h0, h2);
            load_taps(n + 2, h2); load_block(n + 2, tA, phA); step(n + 1, tB, phB, h1, h0);
            load_taps(n + 3, h0); load_block(n + 3, tB, phB); step(n + 2, tA, phA, h2, h1);
            if (n + 3 > 32) break;
            load_taps(n + 4, h1); load_block(n + 4, tA, phA); step(n + 3, tB, phB, h0, h2);
            load_taps(n + 5, h2); load_block(n + 5, tB, phB); step(n + 4, tA, phA, h1, h0);
            load_taps(n + 6, h0); load_block(n + 6, tA, phA); step(n + 5, tB, phB, h2, h1);
        }
        const float wr0 = w0.x, wi1 = w1.y;
        // Iterate(): out[k] for block index mod 4 (qs and T are multiples of 4; o0 is even)
        const float v0 = (o0 & 2) ? -wr0 : wr0;
        const float v1 = (o0 & 2) ? wi1 * cur.sign : -wi1 * cur.sign;
        CWSLG_GLOBAL v2f *out2 = reinterpret_cast<CWSLG_GLOBAL v2f *>(as_global_rw(cur.out) + (size_t)cur.tile * T + o0);
        v2f ov; ov.x = v0; ov.y = v1;
        *out2 = ov;
        float mx = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
        for (int msk = 32; msk >= 1; msk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, msk, 64));
        if ((tid & 63) == 0) publish_peak(cur.peak, mx);
    }
}



// ---- rounds 3 and 4: the exact mode as tiles of one channel (lane = output pair), superseded by demod_exact5_kernel (CWSLG_DEMOD_VARIANT 23-27)
#include "exact3_asm.inc"
#include "exact4_asm.inc"
// ---------------------------------------------------------------------------------------------
// Registers of demod_exact3_kernel's software pipeline, filled by hand-issued loads (see the kernel).
typedef float v8f __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int D>
struct ExactBlock {                 // one block's D mixed samples and its mixer phase
    // One 64-bit register pair per complex sample, filled by ds_read_b64.  (hipcc reaches either word of a 64-bit operand through
    // op_sel but copies the FOURTH word of a 128-bit value into a fresh register before it broadcasts it -- 8 v_mov per step with
    // ds_read_b128 -- and the form that needs no VGPR broadcast needs the same of SGPR quads, 14 s_mov per step.  Measured on one box,
    // 512 slots: 8-byte reads 4.92 ms, 16-byte reads with scalar broadcasts 4.95 ms: the LDS is not what bounds this kernel.)
    v2f q[D];
    v2f php;                        // the block's mixer phase: the row's pad slot
    __device__ __forceinline__ v2f t(int m) const { return q[m]; }
    __device__ __forceinline__ v2f ph() const { return php; }
    __device__ __forceinline__ v2f phn() const { return v2f{-php.y, php.x}; }
};

template <int D> struct ExactTaps;   // one tap row = D wave-uniform pairs (h[m + D n], h[m + D (n-1)]) in SGPRs
template <>
struct ExactTaps<16> {
    v16f lo, hi;
    __device__ __forceinline__ v2f pair(int m) const { return m < 8 ? v2f{lo[2 * m], lo[2 * m + 1]} : v2f{hi[2 * m - 16], hi[2 * m - 15]}; }
};
template <>
struct ExactTaps<8> {
    v16f lo;
    __device__ __forceinline__ v2f pair(int m) const { return v2f{lo[2 * m], lo[2 * m + 1]}; }
};
template <>
struct ExactTaps<4> {
    v8f lo;
    __device__ __forceinline__ v2f pair(int m) const { return v2f{lo[2 * m], lo[2 * m + 1]}; }
};

// Issue every load of one FIR step back to back -- ONE assembly statement, so that hipcc cannot spread them over the step that is
// being computed meanwhile (it did, down to the last third of it: the step's registers die one by one and it reused them in
// place, which left the LDS a fraction of a step to answer): the tap row through the scalar cache, the block's D + 1 LDS
// words of 8 bytes.  Early-clobber outputs: the address operands are read by every instruction of the statement.  `pin` is the first
// sample of the block about to be COMPUTED, passed through untouched: its sums start from it, so the step's arithmetic cannot be
// scheduled ahead of this statement (hipcc otherwise sinks the statement, whose many results lengthen live ranges, below most of it).
// ROFF / TOFF: compile-time byte offsets of the row within the lane's LDS window and of the tap row within the table -- the FIR is
// straight-line code (33 steps unrolled), so neither address ever needs an instruction.
#define X3_RD(i) "ds_read_b64 %" #i ", %[row] offset:%c[ro]+8*" #i "\n\t"
template <int ROFF, int TOFF>
__device__ __forceinline__ void exact_issue(ExactBlock<16> &b, ExactTaps<16> &h, unsigned row_addr, const CWSLG_CONST float *taps, v2f &pin)
{
    asm volatile("s_load_dwordx16 %[tl], %[tp], %c[to]\n\ts_load_dwordx16 %[th], %[tp], %c[to]+0x40\n\t"
                 X3_RD(0) X3_RD(1) X3_RD(2) X3_RD(3) X3_RD(4) X3_RD(5) X3_RD(6) X3_RD(7) X3_RD(8) X3_RD(9) X3_RD(10) X3_RD(11)
                 X3_RD(12) X3_RD(13) X3_RD(14) X3_RD(15) X3_RD(16)
                 : "=&v"(b.q[0]), "=&v"(b.q[1]), "=&v"(b.q[2]), "=&v"(b.q[3]), "=&v"(b.q[4]), "=&v"(b.q[5]), "=&v"(b.q[6]), "=&v"(b.q[7]),
                   "=&v"(b.q[8]), "=&v"(b.q[9]), "=&v"(b.q[10]), "=&v"(b.q[11]), "=&v"(b.q[12]), "=&v"(b.q[13]), "=&v"(b.q[14]),
                   "=&v"(b.q[15]), "=&v"(b.php), [tl] "=&s"(h.lo), [th] "=&s"(h.hi), "+v"(pin)
                 : [row] "v"(row_addr), [tp] "s"(taps), [ro] "n"(ROFF), [to] "n"(TOFF) : "memory");
}
template <int ROFF, int TOFF>
__device__ __forceinline__ void exact_issue(ExactBlock<8> &b, ExactTaps<8> &h, unsigned row_addr, const CWSLG_CONST float *taps, v2f &pin)
{
    asm volatile("s_load_dwordx16 %[tl], %[tp], %c[to]\n\t"
                 X3_RD(0) X3_RD(1) X3_RD(2) X3_RD(3) X3_RD(4) X3_RD(5) X3_RD(6) X3_RD(7) X3_RD(8)
                 : "=&v"(b.q[0]), "=&v"(b.q[1]), "=&v"(b.q[2]), "=&v"(b.q[3]), "=&v"(b.q[4]), "=&v"(b.q[5]), "=&v"(b.q[6]), "=&v"(b.q[7]),
                   "=&v"(b.php), [tl] "=&s"(h.lo), "+v"(pin)
                 : [row] "v"(row_addr), [tp] "s"(taps), [ro] "n"(ROFF), [to] "n"(TOFF) : "memory");
}
template <int ROFF, int TOFF>
__device__ __forceinline__ void exact_issue(ExactBlock<4> &b, ExactTaps<4> &h, unsigned row_addr, const CWSLG_CONST float *taps, v2f &pin)
{
    asm volatile("s_load_dwordx8 %[tl], %[tp], %c[to]\n\t"
                 X3_RD(0) X3_RD(1) X3_RD(2) X3_RD(3) X3_RD(4)
                 : "=&v"(b.q[0]), "=&v"(b.q[1]), "=&v"(b.q[2]), "=&v"(b.q[3]), "=&v"(b.php), [tl] "=&s"(h.lo), "+v"(pin)
                 : [row] "v"(row_addr), [tp] "s"(taps), [ro] "n"(ROFF), [to] "n"(TOFF) : "memory");
}
// D = 16: the wait for step n's loads and the issue of step n + 1's as ONE statement (hipcc pads every statement boundary with an
// s_nop: one issue slot of ~90 per step).  `cur_php`, the only loaded register the compiler's own code reads (the sums are assembly
// statements of their own, kept in order behind this one), is tied so that no such read is scheduled above the wait.
template <int ROFF, int TOFF>
__device__ __forceinline__ void exact_wait_issue(v2f &cur_php, ExactBlock<16> &b, ExactTaps<16> &h, unsigned row_addr, const CWSLG_CONST float *taps, v2f &pin)
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "s_load_dwordx16 %[tl], %[tp], %c[to]\n\ts_load_dwordx16 %[th], %[tp], %c[to]+0x40\n\t"
                 X3_RD(0) X3_RD(1) X3_RD(2) X3_RD(3) X3_RD(4) X3_RD(5) X3_RD(6) X3_RD(7) X3_RD(8) X3_RD(9) X3_RD(10) X3_RD(11)
                 X3_RD(12) X3_RD(13) X3_RD(14) X3_RD(15) X3_RD(16)
                 : "=&v"(b.q[0]), "=&v"(b.q[1]), "=&v"(b.q[2]), "=&v"(b.q[3]), "=&v"(b.q[4]), "=&v"(b.q[5]), "=&v"(b.q[6]), "=&v"(b.q[7]),
                   "=&v"(b.q[8]), "=&v"(b.q[9]), "=&v"(b.q[10]), "=&v"(b.q[11]), "=&v"(b.q[12]), "=&v"(b.q[13]), "=&v"(b.q[14]),
                   "=&v"(b.q[15]), "=&v"(b.php), [tl] "=&s"(h.lo), [th] "=&s"(h.hi), "+v"(pin), "+v"(cur_php)
                 : [row] "v"(row_addr), [tp] "s"(taps), [ro] "n"(ROFF), [to] "n"(TOFF) : "memory");
}
#undef X3_RD
// Wait for everything the wave has in flight on the LDS / scalar-memory counter; the tied operands make every later use of the
// registers the hand-issued loads fill depend on this statement -- and the statement depend on `w`, the running sum of the step
// before: without that hipcc hoists the wait to just behind the loads it covers and sinks the whole step's arithmetic below it.
__device__ __forceinline__ void exact_wait(ExactBlock<16> &b, ExactTaps<16> &h, v2f &w)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3]), "+v"(b.q[4]), "+v"(b.q[5]), "+v"(b.q[6]), "+v"(b.q[7]),
                   "+v"(b.q[8]), "+v"(b.q[9]), "+v"(b.q[10]), "+v"(b.q[11]), "+v"(b.q[12]), "+v"(b.q[13]), "+v"(b.q[14]), "+v"(b.q[15]),
                   "+v"(b.php), "+s"(h.lo), "+s"(h.hi), "+v"(w) :: "memory");
}
__device__ __forceinline__ void exact_wait(ExactBlock<8> &b, ExactTaps<8> &h, v2f &w)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3]), "+v"(b.q[4]), "+v"(b.q[5]), "+v"(b.q[6]), "+v"(b.q[7]),
                   "+v"(b.php), "+s"(h.lo), "+v"(w) :: "memory");
}
__device__ __forceinline__ void exact_wait(ExactBlock<4> &b, ExactTaps<4> &h, v2f &w)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b.q[0]), "+v"(b.q[1]), "+v"(b.q[2]), "+v"(b.q[3]), "+v"(b.php), "+s"(h.lo), "+v"(w) :: "memory");
}

// The state a FIR step leaves for its "tail" -- sum * phase and the accumulation into the workspace slots (:170) -- which is computed
// one step LATER, interleaved with the next step's sums: the tail is a chain of four dependent packed operations (~11 cycles each from
// one wave), and run at the end of its own step it sat, with nothing to overlap it, in front of the next step's wait.
struct ExactTail { v2f sX, sY, ph; };

// D = 16: the 64 packed operations of a step's sums, and the 4 of the previous step's tail, in a HAND-WRITTEN order (two assembly
// statements, samples 0-7 and 8-15).  A wave issues one instruction per ~5 cycles and a dependent packed operation waits ~11
// (scripts/micro/pk_latency.hip), so an operation must sit at least three instructions behind the one it depends on; hipcc's own orders
// of this code put each add right behind its multiply (and an s_nop between them), or one whole chain behind the other.  Here every
// product is made four instructions ahead of the add that consumes it (two product register pairs per chain, alternating), the two
// chains alternate, and the tail's four operations are dropped into the first gaps.  Same IEEE operations on the same operands.
#define X3_MX(P, m) "v_pk_mul_f32 %[" #P "], %[h" #m "], %[q" #m "] op_sel_hi:[1,0]\n\t"      /* (q.x * h.lo, q.x * h.hi) */
#define X3_MY(P, m) "v_pk_mul_f32 %[" #P "], %[h" #m "], %[q" #m "] op_sel:[0,1]\n\t"         /* (q.y * h.lo, q.y * h.hi) */
#define X3_AX(P) "v_pk_add_f32 %[sx], %[sx], %[" #P "]\n\t"
#define X3_AY(P) "v_pk_add_f32 %[sy], %[sy], %[" #P "]\n\t"
#define X3_QH(b, h, m) [q##m] "v"(b.q[m]), [h##m] "s"(h.pair(m))
template <bool TAIL>
__device__ __forceinline__ void exact_sums16_lo(const ExactBlock<16> &b, const ExactTaps<16> &h, v2f &sX, v2f &sY, const ExactTail &prev, v2f &W)
{
    v2f xa, ya, xb, yb, ta, tb;
    if (TAIL) {
        asm volatile(X3_MX(sx, 0) X3_MY(sy, 0)
                     "v_pk_mul_f32 %[ta], %[sxp], %[ph]\n\t"                              /* (ac of o0, ad of o0 + 1) */
                     X3_MX(xb, 1) X3_MY(yb, 1)
                     "v_pk_mul_f32 %[tb], %[syp], %[ph] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"   /* (Im sum_o0 * -ph.y, Im sum_o1 * ph.x) = (-(bd), bc) */
                     X3_MX(xa, 2) X3_MY(ya, 2) X3_AX(xb) X3_AY(yb)
                     "v_pk_add_f32 %[ta], %[ta], %[tb]\n\t"                               /* (ac - bd, ad + bc) */
                     X3_MX(xb, 3) X3_MY(yb, 3) X3_AX(xa) X3_AY(ya)
                     "v_pk_add_f32 %[w], %[w], %[ta]\n\t"                                 /* workspace slots += */
                     X3_MX(xa, 4) X3_MY(ya, 4) X3_AX(xb) X3_AY(yb)
                     X3_MX(xb, 5) X3_MY(yb, 5) X3_AX(xa) X3_AY(ya)
                     X3_MX(xa, 6) X3_MY(ya, 6) X3_AX(xb) X3_AY(yb)
                     X3_MX(xb, 7) X3_MY(yb, 7) X3_AX(xa) X3_AY(ya)
                     X3_AX(xb) X3_AY(yb)
                     : [sx] "=&v"(sX), [sy] "=&v"(sY), [xa] "=&v"(xa), [ya] "=&v"(ya), [xb] "=&v"(xb), [yb] "=&v"(yb), [ta] "=&v"(ta),
                       [tb] "=&v"(tb), [w] "+v"(W)
                     : X3_QH(b, h, 0), X3_QH(b, h, 1), X3_QH(b, h, 2), X3_QH(b, h, 3), X3_QH(b, h, 4), X3_QH(b, h, 5), X3_QH(b, h, 6), X3_QH(b, h, 7),
                       [sxp] "v"(prev.sX), [syp] "v"(prev.sY), [ph] "v"(prev.ph));
    } else {
        asm volatile(X3_MX(sx, 0) X3_MY(sy, 0) X3_MX(xb, 1) X3_MY(yb, 1)
                     X3_MX(xa, 2) X3_MY(ya, 2) X3_AX(xb) X3_AY(yb)
                     X3_MX(xb, 3) X3_MY(yb, 3) X3_AX(xa) X3_AY(ya)
                     X3_MX(xa, 4) X3_MY(ya, 4) X3_AX(xb) X3_AY(yb)
                     X3_MX(xb, 5) X3_MY(yb, 5) X3_AX(xa) X3_AY(ya)
                     X3_MX(xa, 6) X3_MY(ya, 6) X3_AX(xb) X3_AY(yb)
                     X3_MX(xb, 7) X3_MY(yb, 7) X3_AX(xa) X3_AY(ya)
                     X3_AX(xb) X3_AY(yb)
                     : [sx] "=&v"(sX), [sy] "=&v"(sY), [xa] "=&v"(xa), [ya] "=&v"(ya), [xb] "=&v"(xb), [yb] "=&v"(yb)
                     : X3_QH(b, h, 0), X3_QH(b, h, 1), X3_QH(b, h, 2), X3_QH(b, h, 3), X3_QH(b, h, 4), X3_QH(b, h, 5), X3_QH(b, h, 6), X3_QH(b, h, 7));
    }
}
#undef X3_QH
#define X3_QH(b, h, m, k) [q##k] "v"(b.q[m]), [h##k] "s"(h.pair(m))
__device__ __forceinline__ void exact_sums16_hi(const ExactBlock<16> &b, const ExactTaps<16> &h, v2f &sX, v2f &sY)
{
    v2f xa, ya, xb, yb;
    asm volatile(X3_MX(xa, 0) X3_MY(ya, 0) X3_MX(xb, 1) X3_MY(yb, 1) X3_AX(xa) X3_AY(ya)
                 X3_MX(xa, 2) X3_MY(ya, 2) X3_AX(xb) X3_AY(yb)
                 X3_MX(xb, 3) X3_MY(yb, 3) X3_AX(xa) X3_AY(ya)
                 X3_MX(xa, 4) X3_MY(ya, 4) X3_AX(xb) X3_AY(yb)
                 X3_MX(xb, 5) X3_MY(yb, 5) X3_AX(xa) X3_AY(ya)
                 X3_MX(xa, 6) X3_MY(ya, 6) X3_AX(xb) X3_AY(yb)
                 X3_MX(xb, 7) X3_MY(yb, 7) X3_AX(xa) X3_AY(ya)
                 X3_AX(xb) X3_AY(yb)
                 : [sx] "+v"(sX), [sy] "+v"(sY), [xa] "=&v"(xa), [ya] "=&v"(ya), [xb] "=&v"(xb), [yb] "=&v"(yb)
                 : X3_QH(b, h, 8, 0), X3_QH(b, h, 9, 1), X3_QH(b, h, 10, 2), X3_QH(b, h, 11, 3), X3_QH(b, h, 12, 4), X3_QH(b, h, 13, 5),
                   X3_QH(b, h, 14, 6), X3_QH(b, h, 15, 7));
}
#undef X3_QH
#undef X3_MX
#undef X3_MY
#undef X3_AX
#undef X3_AY

// Steps 2 P and 2 P + 1 of demod_exact3_kernel's FIR for P = 0..15 (step 32 follows in the kernel): wait for the step's loads, issue the
// next step's, compute this step's sums (D = 16: with the previous step's tail inside them).  Even steps use buffers A (row P of the
// even-block array), odd steps buffers B (row P of the odd-block array).
template <int D, int ROW, int TROW, int P, typename Sums, typename Tail>
__device__ __forceinline__ void exact3_step_pair(ExactBlock<D> &bA, ExactBlock<D> &bB, ExactTaps<D> &hA, ExactTaps<D> &hB, unsigned lds0, unsigned lds1,
                                                 const CWSLG_CONST float *h2, ExactTail &t, v2f &W, Sums &sums, Tail &tail)
{
    if constexpr (D == 16) {
        exact_wait_issue<P * ROW, (2 * P + 1) * TROW>(bA.php, bB, hB, lds1, h2, t.sY);   // step 2 P: its loads have landed; loads of step 2 P + 1
    } else {
        exact_wait(bA, hA, t.sY);
        exact_issue<P * ROW, (2 * P + 1) * TROW>(bB, hB, lds1, h2, bA.q[0]);
    }
    sums(bA, hA, t, std::integral_constant<bool, (P > 0)>{});                    // ... and the tail of step 2 P - 1
    if constexpr (D == 16) {
        exact_wait_issue<(P + 1) * ROW, (2 * P + 2) * TROW>(bB.php, bA, hA, lds0, h2, t.sY);   // step 2 P + 1; loads of step 2 P + 2
    } else {
        exact_wait(bB, hB, t.sY);
        exact_issue<(P + 1) * ROW, (2 * P + 2) * TROW>(bA, hA, lds0, h2, bB.q[0]);
    }
    if (P == 0) { tail(t, true, false); sums(bB, hB, t, std::false_type{}); }    // step 0's tail: tap block -1 does not exist
    else sums(bB, hB, t, std::true_type{});
}
template <int D, int ROW, int TROW, typename Sums, typename Tail, int... Ps>
__device__ __forceinline__ void exact3_steps(std::integer_sequence<int, Ps...>, ExactBlock<D> &bA, ExactBlock<D> &bB, ExactTaps<D> &hA, ExactTaps<D> &hB,
                                             unsigned lds0, unsigned lds1, const CWSLG_CONST float *h2, ExactTail &t, v2f &W, Sums &sums, Tail &tail)
{
    (exact3_step_pair<D, ROW, TROW, Ps>(bA, bB, hA, hB, lds0, lds1, h2, t, W, sums, tail), ...);
}

// ---------------------------------------------------------------------------------------------
// demod_exact3_kernel: ProcessBlock's arithmetic, operation for operation (bit-identical frames).
//
//   A thread owns outputs o (even) and o + 1.  Iterate() reads only Re of the even output's workspace slot and only Im of the odd
//   one's (SSBD.hpp:131-134), and at step n block o + n feeds BOTH of them -- with tap block n for o, n - 1 for o + 1 -- through the
//   SAME mixed samples t[m] and the same block phase.  So the running sums live transposed:
//       sX = (Re sum_o, Re sum_{o+1}),  sY = (Im sum_o, Im sum_{o+1})
//       sX += (t.x, t.x) * (h[m + D n], h[m + D (n-1)])        one v_pk_mul_f32 + one v_pk_add_f32, un-fused (:167-168)
//       sY += (t.y, t.y) * (same tap pair)
//   and the tap PAIR is wave-uniform: it comes from a host-interleaved table taps2[33][D][2] through the SCALAR cache into an SGPR
//   pair that the packed multiply reads directly -- no vector-memory tap loads (exact2's four broadcast global_load_dwordx4 per step
//   and wave kept the CU's texture-address path about as busy as its VALU), no tap VGPRs, no register shuffles for odd taps.
//   sum * phase (:170) is needed in one component per output only:
//       (Re_o, Im_{o+1}) = sX * (ph.x, ph.y) + sY * (-ph.y, ph.x)        two v_pk_mul_f32 + one v_pk_add_f32, then W += ...
//   (The first product of a block starts the sum instead of being added to 0, which can only change the sign of a zero SUM -- and a
//   workspace slot that starts at +0 and is only ever added to cannot see the sign of a zero addend.)
//   Samples before the demodulator's origin are stored as exact zeros by the mix (wave-uniform slow path, first tiles of a slot
//   only), so the FIR loop carries no per-lane origin test: their blocks contribute +-0 to a slot that is still +0.
//   Steps 0 and 32 touch one output only (tap blocks -1 and 32 do not exist): the other output's addend is replaced by +0 there.
//   The loop's LDS reads and scalar loads are issued BY HAND one step ahead (see the loop), and a workgroup walks a run of tiles with
//   the next tile's HBM loads in flight under its FIR.
template <int D, int T, int NT, bool ASMFIR = true>
__global__ __launch_bounds__(NT, 2) void demod_exact3_kernel(const ChanWork *__restrict__ works,
                                                              const float *__restrict__ taps2,
                                                              int tiles_x, int n_ch, unsigned *__restrict__ xcd_next, int run_len,
                                                              unsigned long long *__restrict__ clk)
{
    using Geo = DemodGeom<D, T>;
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    constexpr int NBH = (Geo::NBLK + 1) / 2 + 1;          // blocks per parity array (+1 slack)
    // Row = the block's D mixed samples + its mixer phase; pitch D + 1 complex = 2 (D + 1) dwords, which is 2 (mod 4): the 32 lanes of
    // a ds_read_b64 group (lane l reads row l + n/2) start on the 32 distinct even banks -- conflict-free.
    // ASMFIR (the product): pitch D + 2 -- 16-byte aligned rows whose start banks (36, 20, 12 dwords = 4 (mod 8) apart) put the eight lanes of a
    // ds_read_b128 group on the 32 banks exactly once -- the whole FIR is one generated assembly statement (exact3_asm.inc) that reads the
    // samples sixteen bytes at a time.  The C++ form (lab library, CWSLG_DEMOD_VARIANT=25) keeps pitch D + 1 and 8-byte reads.
    constexpr int BP = ASMFIR ? D + 2 : D + 1;
    static_assert(2 * NT >= T && T % 4 == 0 && D % 4 == 0, "two outputs per thread");
    __shared__ __attribute__((aligned(16))) float2 s_t[2][NBH * BP];
    static_assert(sizeof(float2) * 2 * NBH * BP <= 81920, "at least two tiles per CU");
#ifdef CWSLG_STAMP
#ifdef CWSLG_STAMP_TOPS
    bool stamp_on = false;
#else
    bool stamp_on = true;              // a workgroup that walks a run of tiles stamps its 100th only
#endif
    int stamp_iter = 0;
#ifndef CWSLG_STAMP_WAVES
    if (threadIdx.x == 0 && blockIdx.x < 65536) {          // where and when the workgroup started
        unsigned hw, xcc;
        unsigned long long t_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)\n\ts_memtime %2\n\ts_waitcnt lgkmcnt(0)"
                     : "=s"(hw), "=s"(xcc), "=s"(t_)::"memory");
        g_stamps[8 * blockIdx.x + 6] = ((unsigned long long)xcc << 32) | hw;
        g_stamps[8 * blockIdx.x + 7] = t_;
    }
#endif
#endif
    STAMP(0);

    // Work items as in demod_kernel: XCD x owns items [x per_xcd, (x + 1) per_xcd), neighbouring items being neighbouring tiles of one
    // channel.  Its workgroups DRAW them from a per-XCD counter (xcd_next[8], zeroed by the launch), one tile ahead: a workgroup that
    // walks a run of tiles keeps the NEXT tile's HBM loads in flight, in registers, under its FIR.  (A fixed share per workgroup was
    // measured 15 % slower than one workgroup per tile: whichever workgroups the dispatcher starts late finish late; a tile here is
    // ~10 us of work, so the counter sees one atomic per ~100 ns.)
    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int lo_item = xcd * per_xcd, hi_item = min((xcd + 1) * per_xcd, total);
    const int tid = threadIdx.x;
    // one draw = a run of run_len consecutive items, chosen by the launch (same-address atomics retire at about one per 100 ns: a
    // draw per tile would bound large launches; small ones use shorter runs so that every CU gets work)
    const int kRun = run_len;
    __shared__ int s_draw[2];          // by draw parity: the draw made during iteration k is read after that iteration's first barrier, while a
                                       // slow wave may still be reading the previous one (run_len == 1: a draw per iteration)
    // (A start offset of half a tile for the CU's second workgroup -- the one in the odd wave slots -- was tried, to put one
    // workgroup's mix and barriers into the other's FIR: no change at 1, 2 or 3 x 8 k cycles.  The tile period is the same 19.8 k ticks
    // from the tenth tile of a run to the last: the workgroups de-phase by themselves, and the FIR is bound by what ONE wave can issue,
    // ~5 cycles per instruction, not by the pipe the two waves of a SIMD share -- scripts/micro/pk_latency.hip.)
    CWSLG_GLOBAL unsigned *ctr = as_global_rw(xcd_next) + xcd;
    if (tid == 0) s_draw[0] = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int item = lo_item + kRun * (int)uni((unsigned)s_draw[0]);
    int draw_par = 1;                                        // the slot the next draw is written to
    if (item >= hi_item) return;
    int run_left = kRun - 1;                                 // items of the current run after `item`
    // The shader clock this launch actually ran at (bench.py's roofline.valu_pipe; MI355X_MICROARCH.md, DVFS item 6: in-kernel clock =
    // delta s_memtime / delta s_memrealtime x 100 MHz).  ONE workgroup of a timed launch (clk != nullptr: cwslg_set_timing) reads the two
    // counters when it has drawn its first run and again when it leaves -- a persistent workgroup lives as long as the launch --
    // and writes them to a host-mapped slot nothing on the device reads.  Untimed launches (clk == nullptr) execute none of it.
    if (clk != nullptr && blockIdx.x == 0 && tid == 0) {
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[0] = t_;
        as_global_rw(clk)[1] = r_;
    }
    TileCtx<D, T> cur;
    int ich, itile;
    item_to_ch_tile(item, tiles_x, n_ch, ich, itile);
    decode_item<D, T>(works + ich, itile, cur);
    v4f xs[NIT];
    float2 ck;
    v4f tn;
    issue_tile_loads<D, T, NT>(cur, tid, xs, ck, tn);
    STAMP(1);
    for (;;) {
    // the run after this one is drawn while its last item is mixed: the atomic's round trip hides under the phasor rebuild and the mix
    unsigned draw = 0;
    if (run_left == 0 && tid == 0) draw = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur.n_out > 0) {
    {
        static_assert(Geo::NCK <= NT, "one checkpoint per lane");
        {
            const int lt = tid;
            const int cidx = cur.ck_first + lt;
            if (lt < Geo::NCK) {
                // blocks before the demodulator's origin (cidx < 0) hold zero samples; their phase slot must still hold a FINITE
                // number (0 * garbage left in LDS by another kernel could be NaN): zero
                float2 p = (cidx >= 0) ? ck : make_float2(0.0f, 0.0f);   // ck: fetched by issue_tile_loads (later items: under the previous FIR)
                const int pbase = cur.pb0 + kCk * lt;
#pragma unroll
                for (int s = 0; s < kCk; ++s) {
                    const int pb = pbase + s;
                    if (pb >= 0 && pb < Geo::NBLK) s_t[pb & 1][(pb >> 1) * BP + D] = p;
                    p = cmul_exact(p, cur.inc);
                }
            }
        }
    }
    STAMP(2);
    // t = in[m] * tone[m]  (SSBD.hpp:167), un-fused, into the parity arrays; x[i < 0] = 0 on the (wave-uniform) slow path
    {
        const float2 tn0 = make_float2(tn.x, tn.y), tn1 = make_float2(tn.z, tn.w);
        const int fv = cur.first_valid;
        auto mix = [&](auto slow_tag) {
            constexpr bool SLOW = decltype(slow_tag)::value;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int r = 2 * tid + it * 2 * NT;
                if (r < Geo::NSAMP) {
                    v4f x = xs[it];
                    if (SLOW) {
                        if (r < fv) x = v4f{0.0f, 0.0f, 0.0f, 0.0f};       // fv is a multiple of D: both samples of the pair
                    }
                    const v2f a = cmul_exact_pk(v2f{x.x, x.y}, v2f{tn0.x, tn0.y});
                    const v2f b = cmul_exact_pk(v2f{x.z, x.w}, v2f{tn1.x, tn1.y});
                    const int blk = r / D, m = r % D;
                    v2f *row = reinterpret_cast<v2f *>(&s_t[blk & 1][(blk >> 1) * BP + m]);   // D = 16: 16-byte aligned (one ds_write_b128); else two ds_write_b64
                    row[0] = a;
                    row[1] = b;
                }
            }
        };
        if (fv != 0) mix(std::true_type{});
        else mix(std::false_type{});
    }
    }   // cur.n_out > 0
    if (run_left == 0 && tid == 0) s_draw[draw_par] = (int)draw;
    lds_barrier();                                           // the tile's LDS image is complete (every load it came from has been consumed)
    STAMP(3);
    // the next item of this workgroup: its IQ, checkpoint and tone loads fly while the FIR below runs (xs, ck, tn are free now)
    const int nitem = run_left ? item + 1 : lo_item + kRun * (int)uni((unsigned)s_draw[draw_par]);
    if (run_left == 0) draw_par ^= 1;
    run_left = run_left ? run_left - 1 : kRun - 1;
    const bool has_next = nitem < hi_item;                   // workgroup-uniform
    TileCtx<D, T> nxt = cur;
    if (has_next) {
        item_to_ch_tile(nitem, tiles_x, n_ch, ich, itile);
        decode_item<D, T>(works + ich, itile, nxt);
        issue_tile_loads<D, T, NT>(nxt, tid, xs, ck, tn);
    }
    const int o0 = 2 * tid;
    float mx_lane = 0.0f;                                    // this lane's |output| maximum (0 for lanes without outputs: the reduction below reads every lane)
    if (o0 < T && o0 < cur.n_out) {
        // The FIR loop's memory operations are issued by hand (inline assembly) so that their ORDER is what is written here: at the
        // top of step n one `s_waitcnt lgkmcnt(0)` covers the LDS reads and the scalar tap loads of step n, which were issued a whole
        // step earlier; then the reads and loads of step n + 1 are issued; then step n is computed.  (Left to hipcc, the loads of every
        // other step sank to their first use -- loop form -- or, fully unrolled, every ds_read was followed by its own wait.)
        // hipcc does not track these operations; the wait statement also "rewrites" every register they fill (tied operands), so no
        // use of a loaded value can be scheduled above it.
        const unsigned lds0 = (unsigned)(uintptr_t)&s_t[0][tid * BP];          // block 2 l of this lane's window
        const unsigned lds1 = (unsigned)(uintptr_t)&s_t[1][tid * BP];          // block 2 l + 1
        const CWSLG_CONST float *h2 = as_const(taps2);
        v2f W = {0.0f, 0.0f};                                // (Re of o0's workspace slot, Im of o0 + 1's): zero after their last read-out (:178)
        if constexpr (ASMFIR) {
            // All 33 steps as ONE assembly statement with every register fixed (scripts/gen_exact3_asm.py, one stream per D): same operations
            // in the same order as the C++ form below; the samples arrive through D / 2 ds_read_b128 per step instead of D + 1 ds_read_b64
            // (inline assembly cannot name the upper pair of a 128-bit operand; fixed registers can).
            static_assert(EXACT3_ASM_ROW_BYTES(D) == BP * (int)sizeof(float2), "exact3_asm.inc is generated for this row pitch");
            if constexpr (D == 16) asm volatile(EXACT3_FIR16_ASM : [w] "+v"(W) : [r0] "v"(lds0), [r1] "v"(lds1), [tp] "s"(h2) : EXACT3_ASM_CLOBBERS_16);
            else if constexpr (D == 8) asm volatile(EXACT3_FIR8_ASM : [w] "+v"(W) : [r0] "v"(lds0), [r1] "v"(lds1), [tp] "s"(h2) : EXACT3_ASM_CLOBBERS_8);
            else asm volatile(EXACT3_FIR4_ASM : [w] "+v"(W) : [r0] "v"(lds0), [r1] "v"(lds1), [tp] "s"(h2) : EXACT3_ASM_CLOBBERS_4);
        } else {
        ExactBlock<D> bA, bB;
        ExactTaps<D> hA, hB;
        auto tail = [&](const ExactTail &t, bool first, bool last) {
            const v2f A = t.sX * t.ph;                       // (ac of o0, ad of o0 + 1)
            const v2f B = t.sY * v2f{-t.ph.y, t.ph.x};       // (-(bd) of o0, bc of o0 + 1): negation commutes with rounding
            v2f R = A + B;                                   // (ac - bd, ad + bc)   (:170)
            if (first) R.y = 0.0f;                           // tap block -1 does not exist
            if (last) R.x = 0.0f;                            // tap block 32 does not exist
            W = W + R;
        };
        // One step's sums; with_tail: the tail of the step before goes first (D = 16: inside the hand-ordered statements).
        // The running sums live TRANSPOSED: sX = (Re sum_o0, Re sum_{o0+1}), sY = (Im sum_o0, Im sum_{o0+1}); the tap pair
        // (h[m + D n], h[m + D (n-1)]) is an SGPR pair the packed multiply reads as it is, the sample word is broadcast (op_sel).
        auto sums = [&](const ExactBlock<D> &b, const ExactTaps<D> &h, ExactTail &t, auto with_tail) {
            constexpr bool TAIL = decltype(with_tail)::value;
            const v2f ph_now = b.ph();
            v2f sX, sY;
            if constexpr (D == 16) {
                exact_sums16_lo<TAIL>(b, h, sX, sY, t, W);
                exact_sums16_hi(b, h, sX, sY);
            } else {
                if (TAIL) tail(t, false, false);
                const v2f t0 = b.t(0);
                sX = v2f{t0.x, t0.x} * h.pair(0);
                sY = v2f{t0.y, t0.y} * h.pair(0);
#pragma unroll
                for (int m = 1; m < D; ++m) {
                    const v2f tm = b.t(m);
                    sX = sX + v2f{tm.x, tm.x} * h.pair(m);   // sr += t.x*h   (:167-168), both outputs
                    sY = sY + v2f{tm.y, tm.y} * h.pair(m);   // si += t.y*h
                }
            }
            t.sX = sX; t.sY = sY; t.ph = ph_now;
        };
        // step n reads block o0 + n = row (n >> 1) of the parity-(n & 1) array relative to this lane's row, and tap row n; the 33 steps
        // are straight-line code (every offset an immediate: no loop counter, no address arithmetic)
        constexpr int ROW = BP * (int)sizeof(float2);       // bytes per LDS row
        constexpr int TROW = 2 * D * (int)sizeof(float);    // bytes per tap row
        ExactTail tl;
        tl.sX = v2f{0.0f, 0.0f}; tl.sY = v2f{0.0f, 0.0f}; tl.ph = v2f{0.0f, 0.0f};
        { v2f none = {0.0f, 0.0f}; exact_issue<0, 0>(bA, hA, lds0, h2, none); }
        exact3_steps<D, ROW, TROW>(std::make_integer_sequence<int, 16>{}, bA, bB, hA, hB, lds0, lds1, h2, tl, W, sums, tail);
        // step 32 (its loads were issued by step 31) with the tail of step 31, then its own tail
        exact_wait(bA, hA, tl.sY);
        sums(bA, hA, tl, std::true_type{});
        tail(tl, false, true);                               // tap block 32 does not exist
        }
        STAMP(4);
#ifdef CWSLG_STAMP_WAVES
        STAMP_WAVE(4);          // slots 4..7: the end of the FIR on waves 0..3 (overwrites slots 5..7 of the other diagnostics)
#endif
        const float wr0 = W.x, wi1 = W.y;
        // Iterate(): out[k] for block index mod 4 (qs and T are multiples of 4; o0 is even)
        const float v0 = (o0 & 2) ? -wr0 : wr0;
        const float v1 = (o0 & 2) ? wi1 * cur.sign : -wi1 * cur.sign;
        CWSLG_GLOBAL v2f *out2 = reinterpret_cast<CWSLG_GLOBAL v2f *>(as_global_rw(cur.out) + (size_t)cur.tile * T + o0);
        v2f ov; ov.x = v0; ov.y = v1;
        *out2 = ov;
        mx_lane = fmaxf(fabsf(v0), fabsf(v1));
    }
    {
        const float mx = wave_max_dpp(mx_lane);              // all 64 lanes take part (DPP and readlane read registers, not exec-masked data)
        if ((tid & 63) == 0) publish_peak(cur.peak, mx);
    }
#ifndef CWSLG_STAMP_WAVES
    STAMP(5);
#endif
    if (!has_next) break;
    lds_barrier();                                           // every wave has finished reading the image the next mix overwrites
    cur = nxt;
    item = nitem;
#ifdef CWSLG_STAMP
#ifdef CWSLG_STAMP_TOPS                  // diagnostic: the loop-top times of eight consecutive tiles (CWSLG_STAMP_TOPS ... + 7), nothing else
    ++stamp_iter;
    if (stamp_iter >= CWSLG_STAMP_TOPS && stamp_iter < CWSLG_STAMP_TOPS + 8 && threadIdx.x == 0 && blockIdx.x < 65536) {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        g_stamps[8 * blockIdx.x + (stamp_iter - CWSLG_STAMP_TOPS)] = t_;
    }
    stamp_on = false;
    if (stamp_iter == CWSLG_STAMP_TOPS + 8) break;
#else
    if (stamp_iter == 100) break;
    ++stamp_iter;
    stamp_on = (stamp_iter == 100);
    STAMP(0); STAMP(1);
#endif
#endif
    }
    if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[2] = t_;
        as_global_rw(clk)[3] = r_;
    }
}

// ---------------------------------------------------------------------------------------------
// demod_exact4_kernel (192 kHz; round 4): demod_exact3_kernel's arithmetic with the 33 steps of an output pair split between TWO waves,
// so that one tile image serves eight waves and a SIMD holds four.
//
//   Why.  exact3 keeps a pair of outputs on one lane for all 33 steps: 250 registers per lane, four waves per 78 KB image, two waves per
//   SIMD.  scripts/micro/pk_issue.hip (profiles/r4_pk_issue.txt): a SIMD retires one packed FP32 operation per 5.2 cycles from one wave,
//   4.46 from two, 4.34 from three, 4.25 from four; exact3 averaged 4.95 (SQ_INSTS_VALU x 4 / GRBM cycles: 81 % pipe-busy) because for
//   a third of a tile's life one of a SIMD's two waves is loading, mixing, waiting at a barrier or storing.  More waves need a smaller
//   footprint PER WAVE, and the image cannot shrink: its rows are the outputs in flight.
//   How.  The only dependence between the steps of an output is the accumulation W = ((0 + R_0) + R_1) + ... + R_32 of the block terms
//   R_n = sum_n * phase_n (SSBD.hpp:170); the R_n are independent.  Waves 0-3 of the workgroup ("A", lane = output pair) run steps 0..16
//   and publish W_A = R_0 + ... + R_16 in LDS; waves 4-7 ("B", the same pairs) run steps 17..32 KEEPING their sixteen R_n in registers,
//   meet the A waves at one barrier, read W_A and finish W = (W_A + R_17) + ... + R_32 -- the reference's additions in the reference's
//   order: bit-identical.  Both streams are generated assembly with every register fixed (exact4_asm.inc, scripts/gen_exact4_asm.py;
//   checked instruction by instruction on the CPU: tests/test_exact4_stream.py) inside 128 registers per lane: the samples of a step
//   live in one 32-register buffer refilled half a step ahead.
//   The workgroup (512 threads) loads, rebuilds the phasor and mixes a tile together (nine 16-byte loads per lane), then FIR, then the
//   next tile: no register prefetch across tiles -- the CU's other workgroup (its four waves per SIMD are two of each) computes meanwhile.
// One stream's whole life (IS_B is a compile-time constant: the two streams keep different registers across their FIR statements, and only
// a per-stream copy of the loop lets the register allocator see that -- with one loop and a run-time test, stream A's prefetch registers
// count as live across stream B's statement and are spilled).  Both copies execute the same barriers in the same order.
template <int T, int NT, bool IS_B>
__device__ __forceinline__ void exact4_stream(const ChanWork *__restrict__ works, const float *__restrict__ taps2, int tiles_x, int n_ch,
                                              unsigned *__restrict__ xcd_next, int run_len, unsigned long long *__restrict__ clk,
                                              float2 *s_tp, v2f *s_w, int *s_draw)
{
    constexpr int D = 16;
    using Geo = DemodGeom<D, T>;
    constexpr int NBH = (Geo::NBLK + 1) / 2 + 1;
    constexpr int BP = D + 2;                              // row pitch in complex samples: exact3's 16-byte-read image
    constexpr int NP = NT / 2;                             // output pairs per tile = lanes per stream
    // The next tile's IQ is prefetched into registers under the FIR, as in exact3 -- but the two streams have different room beside their
    // fixed registers (A: v0-v71, B: v0-v39), so an A lane carries ITA 16-byte loads and a B lane ITB: A covers samples [0, 2 NP ITA),
    // B the rest.
    constexpr int ITA = 13, ITB = 4;
    constexpr int RB0 = 2 * NP * ITA;                      // first sample the B lanes load
    static_assert(2 * NP == T && NT % 128 == 0 && Geo::NCK <= NP, "one lane per output pair and stream; the phasor rebuild runs on stream A's lanes");
    static_assert(RB0 + 2 * NP * ITB >= Geo::NSAMP && RB0 < Geo::NSAMP && RB0 % D == 0, "the two streams' loads cover the tile");
    static_assert(EXACT4_ASM_ROW_BYTES == BP * (int)sizeof(float2), "exact4_asm.inc is generated for this row pitch");
    float2 (*s_t)[NBH * BP] = reinterpret_cast<float2 (*)[NBH * BP]>(s_tp);
    constexpr bool is_b = IS_B;

    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int xcd = blockIdx.x & 7;
    const int lo_item = xcd * per_xcd, hi_item = min((xcd + 1) * per_xcd, total);
    const int kRun = run_len;
    CWSLG_GLOBAL unsigned *ctr = as_global_rw(xcd_next) + xcd;
    if (threadIdx.x == 0) s_draw[0] = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int item = lo_item + kRun * (int)uni((unsigned)s_draw[0]);
    if (item >= hi_item) return;
    int draw_par = 1;
    int run_left = kRun - 1;
    if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {     // the launch's shader clock (see demod_exact3_kernel)
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[0] = t_;
        as_global_rw(clk)[1] = r_;
    }
    const CWSLG_CONST float *h2 = as_const(taps2);
    v4f xs[IS_B ? ITB : ITA];
    float2 ck = make_float2(0.0f, 0.0f);
    v4f tn;
    // this lane's loads of a tile: samples R0 + 2 pair + 2 NP it, it < ITA (stream A) / ITB (stream B); unconditional, the one load that can
    // reach beyond the tile (stream B's last) clamped.  All but one tile per ring revolution lie in one piece: one uniform base + a lane offset.
    auto issue = [&](const TileCtx<D, T> &c, int pair) {
        constexpr int R0 = IS_B ? RB0 : 0, NITS = IS_B ? ITB : ITA;
        const CWSLG_GLOBAL v4f *ring4 = as_global(reinterpret_cast<const v4f *>(c.ring));
        if (c.base + (unsigned)Geo::NSAMP <= c.cap) {
            const CWSLG_GLOBAL v4f *p = ring4 + (c.base >> 1) + (R0 >> 1);
#pragma unroll
            for (int it = 0; it < NITS; ++it) {
                int q = pair + it * NP;                                       // 16-byte index relative to p
                if (R0 + 2 * (NP - 1) + it * 2 * NP > Geo::NSAMP - 2) q = min(q, (Geo::NSAMP - 2 - R0) >> 1);   // (compile-time: stream B's last load only)
                xs[it] = p[q];
            }
        } else {
#pragma unroll
            for (int it = 0; it < NITS; ++it) {
                int r = R0 + 2 * pair + it * 2 * NP;
                if (r > Geo::NSAMP - 2) r = Geo::NSAMP - 2;
                unsigned idx = c.base + (unsigned)r;
                if (idx >= c.cap) idx -= c.cap;
                xs[it] = ring4[idx >> 1];
            }
        }
        if constexpr (!IS_B) {
            int cidx = c.ck_first + ((pair < Geo::NCK) ? pair : 0);
            if (cidx < 0) cidx = 0;
            const v2f t = as_global(reinterpret_cast<const v2f *>(c.ckpt))[cidx];
            ck = make_float2(t.x, t.y);
        }
        tn = as_global(reinterpret_cast<const v4f *>(c.tone))[((2 * pair) % D) >> 1];   // tone[m0], tone[m0+1]
    };
    TileCtx<D, T> cur;
    {
        int ich, itile;
        item_to_ch_tile(item, tiles_x, n_ch, ich, itile);
        decode_item<D, T>(works + ich, itile, cur);
        issue(cur, (int)(threadIdx.x & (NP - 1)));
    }
    for (;;) {
        // An opaque copy of the thread index per tile: hipcc would otherwise hoist every per-lane address of the loads and of the mix's LDS
        // writes out of the loop, and values that live across the FIR statements (which own v40-v127) are spilled to scratch.
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int pair = tid & (NP - 1);
        // the run after this one is drawn while its last item is mixed
        unsigned draw = 0;
        if (run_left == 0 && tid == 0) draw = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur.n_out > 0) {
            if constexpr (!IS_B) {
                const int cidx = cur.ck_first + pair;
                if (pair < Geo::NCK) {
                    float2 p = (cidx >= 0) ? ck : make_float2(0.0f, 0.0f);   // blocks before the origin: a FINITE phase (their samples are zero)
                    const int pbase = cur.pb0 + kCk * pair;
#pragma unroll
                    for (int s = 0; s < kCk; ++s) {
                        const int pb = pbase + s;
                        if (pb >= 0 && pb < Geo::NBLK) s_t[pb & 1][(pb >> 1) * BP + D] = p;
                        p = cmul_exact(p, cur.inc);
                    }
                }
            }
            // t = in[m] * tone[m]  (SSBD.hpp:167), un-fused, into the parity arrays; x[i < 0] = 0 on the (wave-uniform) slow path.
            // A lane's loads are 2 NP samples = 2 NP / D blocks apart, an EVEN number of blocks: all of them land in the same parity array,
            // 2 NP / (2 D) rows apart -- one LDS address per lane and an immediate offset per load (the generic r / D, r % D arithmetic
            // cost ~90 VALU instructions per wave and tile: 7 % of the kernel's, at the power limit 7 % of its energy).
            {
                const float2 tn0 = make_float2(tn.x, tn.y), tn1 = make_float2(tn.z, tn.w);
                const int fv = cur.first_valid;
                constexpr int R0 = IS_B ? RB0 : 0, NITS = IS_B ? ITB : ITA;
                constexpr int ROWSTEP = (2 * NP) / (2 * D);                      // rows of one parity array between consecutive loads
                static_assert((2 * NP) % (2 * D) == 0 && R0 % (2 * D) == 0, "a lane's samples stay in one parity array");
                const int blk0 = R0 / D + (pair >> 3);                            // block of the lane's first sample pair (D = 16: eight lanes per block)
                v2f *row0 = reinterpret_cast<v2f *>(&s_t[blk0 & 1][(blk0 >> 1) * BP + ((2 * pair) & (D - 1))]);
                const int r0 = R0 + 2 * pair;
                auto mix = [&](auto slow_tag) {
                    constexpr bool SLOW = decltype(slow_tag)::value;
#pragma unroll
                    for (int it = 0; it < NITS; ++it) {
                        const int r = r0 + it * 2 * NP;
                        if (R0 + 2 * (NP - 1) + it * 2 * NP >= Geo::NSAMP && r >= Geo::NSAMP) continue;     // only the last load of stream B can lie beyond the tile
                        v4f x = xs[it];
                        if (SLOW) {
                            if (r < fv) x = v4f{0.0f, 0.0f, 0.0f, 0.0f};       // fv is a multiple of D: both samples of the pair
                        }
                        const v2f a = cmul_exact_pk(v2f{x.x, x.y}, v2f{tn0.x, tn0.y});
                        const v2f b = cmul_exact_pk(v2f{x.z, x.w}, v2f{tn1.x, tn1.y});
                        v2f *row = row0 + it * ROWSTEP * BP;                     // 16-byte aligned: one ds_write_b128
                        row[0] = a;
                        row[1] = b;
                    }
                };
                if (fv != 0) mix(std::true_type{});
                else mix(std::false_type{});
            }
        }
        if (run_left == 0 && tid == 0) s_draw[draw_par] = (int)draw;
        lds_barrier();                                       // the tile's image is complete (every load it came from has been consumed)
        // the next item of this workgroup: its loads fly while the FIR below runs (xs, ck, tn are free now)
        const int nitem = run_left ? item + 1 : lo_item + kRun * (int)uni((unsigned)s_draw[draw_par]);
        if (run_left == 0) draw_par ^= 1;
        run_left = run_left ? run_left - 1 : kRun - 1;
        const bool has_next = nitem < hi_item;               // workgroup-uniform
        TileCtx<D, T> nxt = cur;
        if (has_next) {
            int ich, itile;
            item_to_ch_tile(nitem, tiles_x, n_ch, ich, itile);
            decode_item<D, T>(works + ich, itile, nxt);
            issue(nxt, pair);
        }
        if (cur.n_out > 0) {
            // Every wave runs its stream whether or not its lanes hold outputs of a ragged last tile (their rows hold the ring's next samples:
            // finite or not, nothing of them is stored): stream B contains the workgroup barrier, which every wave must reach exactly once.
            const unsigned lds0 = (unsigned)(uintptr_t)&s_t[0][pair * BP];          // block 2 l of this pair's window
            const unsigned lds1 = (unsigned)(uintptr_t)&s_t[1][pair * BP];          // block 2 l + 1
            if constexpr (!IS_B) {
                v2f W = {0.0f, 0.0f};
                asm volatile(EXACT4_FIRA_ASM : [w] "+v"(W) : [r0] "v"(lds0), [r1] "v"(lds1), [tp] "s"(h2) : EXACT4_ASM_CLOBBERS_A);
                s_w[pair] = W;
                lds_barrier();                               // W_A is published; every FIR read of the image is done (stream B: inside its statement)
            } else {
                const unsigned xaddr = (unsigned)(uintptr_t)&s_w[pair];
                v2f W;
                asm volatile(EXACT4_FIRB_ASM : [w] "=&v"(W) : [r0] "v"(lds0), [r1] "v"(lds1), [tp] "s"(h2), [xa] "v"(xaddr) : EXACT4_ASM_CLOBBERS_B);
                const int o0 = 2 * pair;
                float mx_lane = 0.0f;
                if (o0 < cur.n_out) {
                    // Iterate(): out[k] for block index mod 4 (qs and T are multiples of 4; o0 is even)
                    const float v0 = (o0 & 2) ? -W.x : W.x;
                    const float v1 = (o0 & 2) ? W.y * cur.sign : -W.y * cur.sign;
                    CWSLG_GLOBAL v2f *out2 = reinterpret_cast<CWSLG_GLOBAL v2f *>(as_global_rw(cur.out) + (size_t)cur.tile * T + o0);
                    v2f ov; ov.x = v0; ov.y = v1;
                    *out2 = ov;
                    mx_lane = fmaxf(fabsf(v0), fabsf(v1));
                }
                const float mx = wave_max_dpp(mx_lane);
                if ((tid & 63) == 0) publish_peak(cur.peak, mx);
            }
        }
        if (!has_next) break;
        cur = nxt;
        item = nitem;
    }
    if (clk != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[2] = t_;
        as_global_rw(clk)[3] = r_;
    }
}

template <int T, int NT>
__global__ __launch_bounds__(NT, 4) void demod_exact4_kernel(const ChanWork *__restrict__ works,
                                                              const float *__restrict__ taps2,
                                                              int tiles_x, int n_ch, unsigned *__restrict__ xcd_next, int run_len,
                                                              unsigned long long *__restrict__ clk)
{
    using Geo = DemodGeom<16, T>;
    constexpr int NBH = (Geo::NBLK + 1) / 2 + 1, BP = 16 + 2, NP = NT / 2;
    __shared__ __attribute__((aligned(16))) float2 s_t[2][NBH * BP];
    __shared__ __attribute__((aligned(8))) v2f s_w[NP];    // W_A of every pair: stream A -> stream B
    __shared__ int s_draw[2];
    static_assert(2 * (sizeof(float2) * 2 * NBH * BP + sizeof(v2f) * NP + 64) <= 163840, "two workgroups per CU");
    // waves 0 .. NP/64 - 1 run stream A, the others stream B (wave-uniform: NP is a multiple of 64)
    if (uni((unsigned)(threadIdx.x >= NP)) == 0) exact4_stream<T, NT, false>(works, taps2, tiles_x, n_ch, xcd_next, run_len, clk, &s_t[0][0], s_w, s_draw);
    else exact4_stream<T, NT, true>(works, taps2, tiles_x, n_ch, xcd_next, run_len, clk, &s_t[0][0], s_w, s_draw);
}


} // namespace cwslg

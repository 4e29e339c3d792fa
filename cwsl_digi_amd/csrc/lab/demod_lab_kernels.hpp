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


} // namespace cwslg

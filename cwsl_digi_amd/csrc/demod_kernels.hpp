// demod_kernels.hpp -- hand-written gfx950 kernels for the CWSL_DIGI per-channel DSP chain.
//
// What is computed (reference: source/SSBD.hpp:127-183, closed form in SURVEY.md 8a-note):
//
//   z_b      = sum_{t=0}^{32D-1} x[D(b-31)+t] * nco[D(b-31)+t] * h[t]       D = Fs/12000, x[i<0] = 0
//   nco[i]   = tone[i mod D] * phase_{i div D},  phase_{q+1} = fl(phase_q * phase_inc)   (float32, UNfused)
//   audio[b] = Re( z_b * j^(b mod 4) )                                       (USB; LSB flips the Im terms)
//
// The reference evaluates this as a recursive overlap-add over 32 workspace slots, one scalar
// thread per channel.  Here it is inverted into a stride-2D polyphase FIR:
//
//   outputs of equal parity p = b mod 2 are 2D input samples apart and need only ONE real
//   component of y = x*nco (Re for even b, Im for odd b).  With t = 2D*v + u  (u < 2D, v < 16):
//
//       audio[qs + 2w + p] = s(w,p) * sum_u sum_v  P_p[u][w+v] * H[u][v],   H[u][v] = h[2D*v+u]
//       P_p[u][w] = component_p( y[ D*(qs-31+p) + 2D*w + u ] )
//
//   i.e. 2D independent 16-tap FIRs ("branches") along w, summed over the branches.
//
// Kernel layout (one workgroup = one tile of T outputs of one channel):
//   phase 0  threads 0..NCK-1 rebuild the bit-exact float32 phasor for the tile's T+31 blocks from
//            per-channel checkpoints (every kCk = 4 blocks) into LDS -- <=4 serial UNfused complex
//            multiplies each, the same rounding sequence as SSBD.hpp:174.
//   phase 1  coalesced 16-B loads of the IQ ring (HBM), complex mix, scatter of Re/Im into the two
//            branch-major LDS planes P_0/P_1.
//   phase 2  lane = branch u: each lane runs its 16-tap FIR over 16 consecutive w (31 LDS floats,
//            256 FMAs, taps in registers), then a halving butterfly over the 2D lanes of a group
//            sums the branches.  Results are staged in LDS and stored as whole rows.
//   epilogue wave shuffle max|audio| -> one atomicMax per wave into the frame's peak word
//            (feeds prepareAudio's normalisation, Instance.cpp:294-316, without a second pass).
//
// MFMA is deliberately not used: per-channel phasors make the contraction channel-specific, the
// f32 MFMA rate equals the VALU rate on gfx950, and the path is HBM-bound (8 B in per sample).
//
// This translation unit is compiled with -ffp-contract=off; every fused multiply-add below is an
// explicit __builtin_fmaf, every bit-exact sequence is plain * and +/-.
#pragma once
#include <type_traits>
#include <utility>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cwslg {

// Pointers fetched from a descriptor in memory are generic to the compiler (flat_load); these are
// known to be HBM addresses, so say so and get global_load / global_store.
#ifndef CWSLG_BF16_DIAG
#define CWSLG_BF16_DIAG 0
#endif
// -DCWSLG_FIR_PK=1: the FIR of demod_kernel on v_pk_fma_f32 (half the issue slots, same FLOPs; measured alternative, see phase 2)
#ifndef CWSLG_FIR_PK
#define CWSLG_FIR_PK 0
#endif
// -DCWSLG_DIAG_NOHALO=1 (timing only, results are garbage; scripts/gpu_nohalo.sh): the tile kernels neither load nor mix the first
// 2 NT samples of a tile -- about the 31-block halo a workgroup that STREAMED a channel's tiles would still hold in LDS -- to bound
// from above what such a streaming form could gain.
#ifndef CWSLG_DIAG_NOHALO
#define CWSLG_DIAG_NOHALO 0
#endif
#ifndef CWSLG_DIAG_LDS
#define CWSLG_DIAG_LDS 0           // 1..3: timing-only diagnostic builds of demod_kernel without one of its LDS phases (scripts/gpu_r4_ldsdiag.sh)
#endif
#define CWSLG_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const CWSLG_GLOBAL T *as_global(const T *p)
{
    return (const CWSLG_GLOBAL T *)(uintptr_t)p;
}
// Constant address space: a uniform load through it is always a scalar load (s_load), whatever stores the kernel makes elsewhere.
// Only for memory no kernel writes (the tap tables).
#define CWSLG_CONST __attribute__((address_space(4)))
template <typename T>
__device__ __forceinline__ const CWSLG_CONST T *as_const(const T *p)
{
    return (const CWSLG_CONST T *)(uintptr_t)p;
}
template <typename T>
__device__ __forceinline__ CWSLG_GLOBAL T *as_global_rw(T *p)
{
    return (CWSLG_GLOBAL T *)(uintptr_t)p;
}

// Blocks between phasor checkpoints.  4 keeps the per-tile serial rebuild at <=4 dependent complex multiplies
// (it sits at the head of every tile) for 1.6 % extra HBM traffic: 8 B per 4 blocks = per 512 B of IQ at 192 kHz.
constexpr int kCk = 4;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
// A complex value at an HBM address taken from a launch descriptor, through the global address space (global_load / global_store: a generic
// pointer makes a flat_ access, which also counts in lgkmcnt and waits with the LDS traffic).  HIP's float2 is a struct and cannot be copied
// through an address-space-qualified pointer, hence the detour over the 8-byte vector type.
__device__ __forceinline__ float2 gld2(const float2 *p) { const v2f t = *as_global(reinterpret_cast<const v2f *>(p)); return make_float2(t.x, t.y); }
__device__ __forceinline__ void gst2(float2 *p, float2 v) { *as_global_rw(reinterpret_cast<v2f *>(p)) = v2f{v.x, v.y}; }

// One entry per active channel per demod launch.
struct alignas(16) ChanWork {
    const float2 *ring;       // receiver IQ ring in HBM (interleaved re,im)
    float        *out;        // &frame[fill] : where the first pending output goes
    unsigned     *peak;       // max|audio| of the frame, as float bits (atomicMax)
    const float2 *ckpt;       // phasor checkpoints: ckpt[c] = phase_{kCk c}
    const float2 *tone;       // tone[D]
    unsigned      ring_cap;   // ring capacity in complex samples
    unsigned      n_blocks;   // pending outputs (= pending samples / D)
    float2        inc;        // phase_inc
    float         sign;       // +1 USB, -1 LSB
    unsigned      lo_mod;     // ring index of the first pending sample (host-computed: no 64-bit division on the device)
    long long     q_first;    // block index of the first pending output, counted from the demodulator's (re)creation
};
static_assert(sizeof(ChanWork) == 80, "descriptor layout (kDescWords lanes stage it through LDS)");

// One entry per channel per finalize launch.
struct alignas(16) FinWork {
    const float *frame;       // finished float frame
    int16_t     *out;         // int16 frame (frame_len samples)
    unsigned    *peak;        // peak word of the finished frame
    unsigned    *peak_next;   // peak word of the frame that becomes the write frame (reset to 0)
    float       *factor_out;  // where the scale factor is published
    float        scale;       // ftaudioscalefactor or wspraudioscalefactor
    unsigned     n_valid;     // samples demodulated into the frame
    unsigned     frame_len;   // 12000*(period+5)
    int          emit;        // 0: discarded frame (startEpochTime == 0) -> only the reset
};

struct alignas(16) PhasorJob {
    float2  *ckpt;
    float2   inc;
    float2   start;       // checkpoint 0: (1, 0) for a fresh demodulator (SSBD.hpp:121), the live phase after Tune(reset = false)
    unsigned n_ckpt;
    unsigned pad_;
};

// One channel's outputs right after SSBD::Tune(F, isUSB, reset = false) (SSBD.hpp:97-123 with :116-121 skipped): the workspace
// keeps the partial sums of the last 31 blocks, which were mixed with the OLD tone and multiplied by the OLD phasor sequence, so
// output b0 + o (b0 = the block at which the retune took effect) sums tap block n over block b0 + o - 31 + n with the old
// parameters where that block precedes b0 and the new ones from b0 on.  Everything is passed by value: the old tuning's tables
// may be gone by the time the samples arrive.
struct alignas(16) TransWork {
    const float2 *ring;
    float        *out;           // output b0 + o_first goes to out[0]
    unsigned     *peak;
    float2        tone_old[16], tone_new[16];
    float2        phase_old[32];  // blocks b0 - 32 .. b0 - 1
    float2        phase_new[32];  // blocks b0 .. b0 + 31
    float         sign;           // the NEW sideband sign: Iterate reads it when the output leaves (SSBD.hpp:131-134)
    unsigned      ring_cap;
    unsigned      pos_b0;         // ring index of the first sample of block b0
    int           o_first, n_out; // outputs b0 + o_first .. b0 + o_first + n_out - 1, all below b0 + 32
    int           blocks_before;  // blocks that exist before b0 since the demodulator's origin (older ones are x = 0), at most 32
};

// ---------------------------------------------------------------------------------------------
// Exact float32 complex product, unfused: the sequence g++ emits for std::complex<float>
// operator* (SSBD.hpp:174 `phase *= phase_inc`).  The TU is built with -ffp-contract=off.
__device__ __forceinline__ float2 cmul_exact(float2 a, float2 b)
{
    const float ac = a.x * b.x;
    const float bd = a.y * b.y;
    const float ad = a.x * b.y;
    const float bc = a.y * b.x;
    return make_float2(ac - bd, ad + bc);
}
// The same four products, the same difference and sum (each rounded on its own: un-fused), as THREE packed operations: (ac, ad),
// (bd, bc), then (ac - bd, ad + bc) with the first component's addend negated (a + (-b) is a - b exactly).  Half the issue slots
// of the six scalar operations; used where what bounds the code is what one wave can issue (demod_exact3_kernel's mix).
__device__ __forceinline__ v2f cmul_exact_pk(v2f a, v2f b)
{
    v2f p1, p2, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p1) : "v"(a), "v"(b));                 // (a.x b.x, a.x b.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0]" : "=v"(p2) : "v"(a), "v"(b));    // (a.y b.y, a.y b.x)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p1), "v"(p2));                    // (ac - bd, ad + bc)
    return r;
}

// Phasor checkpoint table, built once per distinct tuning at channel-open time.  The recurrence is serial, so
// pass 1 walks it with one lane per channel, keeping only every kCoarse-th checkpoint; pass 2 fills the checkpoints in
// between with one lane per (channel, coarse segment), restarting from the coarse value -- the same multiplies in
// the same order, so the table is bit-identical to a single serial walk (tests/test_gpu_demod.py).
constexpr int kCoarse = 256;            // fine checkpoints per coarse segment (= 1024 blocks)

__global__ void phasor_coarse_kernel(const PhasorJob *__restrict__ jobs, int n_jobs)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_jobs) return;
    const PhasorJob job = jobs[j];
    float2 p = job.start;                               // (1, 0): SSBD.hpp:121
    for (unsigned c = 0; c < job.n_ckpt; c += kCoarse) {
        gst2(job.ckpt + c, p);
        for (int s = 0; s < kCoarse * kCk; ++s) p = cmul_exact(p, job.inc);
    }
}

// grid (ceil(max segments / 64), n_jobs), 64 threads: lane = coarse segment
__global__ void phasor_fine_kernel(const PhasorJob *__restrict__ jobs)
{
    const PhasorJob job = jobs[blockIdx.y];
    const unsigned seg = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned c0 = seg * kCoarse;
    if (c0 >= job.n_ckpt) return;
    float2 p = gld2(job.ckpt + c0);
    for (unsigned c = c0 + 1; c < c0 + kCoarse && c < job.n_ckpt; ++c) {
#pragma unroll
        for (int s = 0; s < kCk; ++s) p = cmul_exact(p, job.inc);
        gst2(job.ckpt + c, p);
    }
}

// ---------------------------------------------------------------------------------------------
// Cross-lane primitives of the branch reduction: no LDS traffic, VALU only.
//   DPP_ROR8        lane i <- lane i^8   (rotate the 16-lane row by 8)
//   DPP_HALF_MIRROR lane i <- lane i^7   (reverse each 8-lane half row: pairs bit2 = 0 with bit2 = 1)
//   DPP_XOR2/XOR1   quad permutes [2,3,0,1] / [1,0,3,2]
constexpr int DPP_ROR8 = 0x128, DPP_HALF_MIRROR = 0x141, DPP_XOR2 = 0x4E, DPP_XOR1 = 0xB1;

template <int CTRL>
__device__ __forceinline__ float dpp_get(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), CTRL, 0xf, 0xf, false));
}
// wavefront maximum of a non-negative float: four DPP steps inside the 16-lane rows, four readlanes across them (a __shfl_xor butterfly
// is six dependent ds_bpermute round trips)
__device__ __forceinline__ float wave_max_dpp(float v)
{
    v = fmaxf(v, dpp_get<DPP_XOR1>(v));
    v = fmaxf(v, dpp_get<DPP_XOR2>(v));
    v = fmaxf(v, dpp_get<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_get<0x140>(v));                            // row_mirror: every lane of a row holds the row maximum
    const int vi = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(vi, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(vi, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(vi, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(vi, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// One halving step over the lane bit BIT (pairing CTRL): lanes with the bit clear keep lo and get the
// partner's lo, lanes with it set keep hi and get the partner's hi.
template <int CTRL>
__device__ __forceinline__ float halve_dpp(float lo, float hi, bool bit_set)
{
    const float keep = bit_set ? hi : lo;
    const float send = bit_set ? lo : hi;
    return keep + dpp_get<CTRL>(send);
}

// Sum acc[0..15] over the GL (16, 8 or 4) lanes of a group by halving: after the step on lane bit B the lanes with
// the bit clear hold the lower half of the remaining outputs, the others the upper half.  On return lane k holds
// NV = 16/GL results acc[0..NV-1] for output indices rbase .. rbase+NV-1.
template <int GL>
__device__ __forceinline__ int reduce_lanes(float (&acc)[16], int k)
{
    static_assert(GL == 16 || GL == 8 || GL == 4, "group width");
    int rbase = 0;
    if constexpr (GL >= 16) {
        const bool b = (k & 8) != 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) acc[r] = halve_dpp<DPP_ROR8>(acc[r], acc[r + 8], b);
        if (b) rbase += 8;
    }
    if constexpr (GL >= 8) {
        const bool b = (k & 4) != 0;
        constexpr int N = (GL == 16) ? 8 : 16;
#pragma unroll
        for (int r = 0; r < N / 2; ++r) acc[r] = halve_dpp<DPP_HALF_MIRROR>(acc[r], acc[r + N / 2], b);
        if (b) rbase += N / 2;
    }
    {
        const bool b = (k & 2) != 0;
        constexpr int N = (GL == 16) ? 4 : (GL == 8 ? 8 : 16);
#pragma unroll
        for (int r = 0; r < N / 2; ++r) acc[r] = halve_dpp<DPP_XOR2>(acc[r], acc[r + N / 2], b);
        if (b) rbase += N / 2;
    }
    {
        const bool b = (k & 1) != 0;
        constexpr int N = (GL == 16) ? 2 : (GL == 8 ? 4 : 8);
#pragma unroll
        for (int r = 0; r < N / 2; ++r) acc[r] = halve_dpp<DPP_XOR1>(acc[r], acc[r + N / 2], b);
        if (b) rbase += N / 2;
    }
    return rbase;
}

// Frame peak: max|audio| as float bits, one word per channel, one fire-and-forget atomicMax per wave.  (Measured: compiled
// out, the tile kernel gains 0.07 ms of 2.75; guarding the atomic with a device-scope load of the word -- to skip it
// when it cannot raise the maximum -- costs more than it saves, 2.85 ms: the load's round trip lands on the workgroup's tail.)
__device__ __forceinline__ void publish_peak(unsigned *peak, float mx)
{
    // through the GLOBAL address space: a generic-pointer atomic is a FLAT instruction, which is counted in lgkmcnt as well as vmcnt
    // and so stalls the next LDS wait for an HBM round trip
    if (mx > 0.0f)
        __hip_atomic_fetch_max((CWSLG_GLOBAL unsigned *)(uintptr_t)peak, __float_as_uint(mx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------
template <int D, int T>
struct DemodGeom {
    static constexpr int G      = 2 * D;              // polyphase branches
    static constexpr int GL     = D;                  // lanes per group: each lane owns the branch PAIR (2k, 2k+1)
    static constexpr int NBLK   = T + 31;             // input blocks per tile
    static constexpr int NSAMP  = D * NBLK;           // input samples per tile
    static constexpr int NW     = T / 2 + 16;         // columns per plane (T/2+15 used, +1 read slack)
    // LDS image of one plane: D pair-rows; row k holds branches 2k and 2k+1 interleaved per column:
    //   float index = k*PR + 2*w + (u & 1)
    // so the two samples a lane mixes per load (branches uu, uu+1, same column) are ONE ds_write_b64, and the FIR lane
    // reads its pair-row as contiguous ds_read_b128.  PR/4 odd: the 16 lanes of a b128 group hit 16 distinct 16-B
    // slots, and the 16 lanes of a b64 write spread over 8 bank pairs (2-way; the [2D][148] image was 4-way on b32).
    static constexpr int PR0    = (2 * NW + 3) / 4 * 4;
    static constexpr int PR     = ((PR0 / 4) % 2 == 1) ? PR0 : PR0 + 4;
    static constexpr int NCK    = (NBLK + kCk - 1) / kCk + 1;   // checkpoints touched by a tile
    static constexpr int PLANE_FLOATS = D * PR;
    static constexpr int AUX_FLOATS   = (2 * (NBLK + 1) > T) ? 2 * (NBLK + 1) : T;  // phases, later the output row
    static constexpr int LDS_BYTES    = (2 * PLANE_FLOATS + AUX_FLOATS) * 4;
};


// ---------------------------------------------------------------------------------------------
// demod_kernel<D, T, NT, MODE>: one tile = T outputs of one channel (see the file header for the phases).
//   MODE 2           persistent workgroups walking a run of work items like MODE 1 but WITHOUT the prefetch (loads issued
//                    at the top of each item): saves the per-workgroup launch, descriptor and tap loads.
//   MODE 0 (PERSIST = false)  one workgroup per (channel, tile) work item; 4 workgroups per CU hide each other's HBM latency.
//                    This is the default: 2.53-2.60 ms per 512-slot launch (85 VGPRs, 40.3 KB of LDS).
//   PERSIST = true   a persistent workgroup walks a run of work items and keeps the NEXT item's HBM loads in
//                    flight (in place: each load register is refilled right after phase 1 has consumed it, 128 VGPRs,
//                    still 4 workgroups per CU) while it computes the current one.  Kept as a measured alternative
//                    (2.89 against 2.75 ms on the same box): the kernel is bounded by VALU issue, LDS
//                    and HBM together at a power-limited clock, not by exposed latency.
// Barriers are raw `s_barrier` behind an `s_waitcnt lgkmcnt(0)`: __syncthreads() would also drain vmcnt, i.e.
// wait for the prefetch.
// Diagnostic build only (-DCWSLG_STAMP, scripts/gpu_stamps.py): per-workgroup s_memtime stamps at the phase seams,
// written to a buffer nothing else reads; the shipped library contains none of this.
#ifdef CWSLG_STAMP
__device__ unsigned long long g_stamps[8 * 65536];
#define STAMP(slot)                                                                                 \
    do {                                                                                            \
        if (threadIdx.x == 0 && blockIdx.x < 65536 && stamp_on) {                                               \
            unsigned long long t_;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            g_stamps[8 * blockIdx.x + (slot)] = t_;                                                 \
        }                                                                                           \
    } while (0)
// ... by lane 0 of EVERY wave, into slot base + wave (how far apart the waves of a workgroup reach a point)
#define STAMP_WAVE(base)                                                                            \
    do {                                                                                            \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 65536 && stamp_on) {                            \
            unsigned long long t_;                                                                  \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");              \
            g_stamps[8 * blockIdx.x + (base) + (threadIdx.x >> 6)] = t_;                            \
        }                                                                                           \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#define STAMP_WAVE(base) do { } while (0)
#endif

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int kDescWords = sizeof(ChanWork) / 4;

// Work-item order.  tiles_x > 0: channel-major (item = ch*tiles_x + tile): neighbouring items are neighbouring tiles of
// one channel, whose 31-block halo then hits the XCD's L2 -- right for private IQ streams.  tiles_x < 0: tile-major
// (item = tile*n_ch + ch): neighbouring items are the channels of one receiver at the same tile, so the receiver's IQ
// is fetched from HBM once and served from L2 to its other channels -- right for the reference's real topology
// (many decoders per band).
__device__ __forceinline__ void item_to_ch_tile(int item, int tiles_x, int n_ch, int &ch, int &tile)
{
    if (tiles_x > 0) { ch = item / tiles_x; tile = item - ch * tiles_x; }
    else { tile = item / n_ch; ch = item - tile * n_ch; }
}

template <int D, int T>
struct TileCtx {                 // wave-uniform description of one work item (held in SGPRs)
    const float2 *ring, *ckpt, *tone;
    float *out;
    unsigned *peak;
    float2 inc;
    float sign;
    int tile, n_out;             // n_out == 0: nothing to do (ragged tail of a channel with fewer pending blocks)
    int first_valid;             // samples r < first_valid precede the demodulator's origin (x[i<0] = 0)
    unsigned base, cap;          // ring index of tile sample r = 0, ring capacity
    int ck_first;                // checkpoint index of lane 0 in phase 0 (may be negative)
    int pb0;                     // s_phase index of block 16*ck_first
};

__device__ __forceinline__ unsigned uni(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
template <typename P>
__device__ __forceinline__ P *uni_ptr(P *p)
{
    const unsigned long long v = (unsigned long long)p;
    return (P *)(((unsigned long long)uni((unsigned)(v >> 32)) << 32) | uni((unsigned)v));
}

// The descriptor was staged in LDS by the workgroup (never read from global memory inside the tile loop: a
// vector-memory read there would force s_waitcnt vmcnt(0) and drain the prefetch).
template <int D, int T>
__device__ __forceinline__ void decode_item(const ChanWork *sd, int tile, TileCtx<D, T> &c)
{
    using Geo = DemodGeom<D, T>;
    c.ring = uni_ptr(sd->ring); c.ckpt = uni_ptr(sd->ckpt); c.tone = uni_ptr(sd->tone);
    c.out = uni_ptr(sd->out); c.peak = uni_ptr(sd->peak);
    c.inc = make_float2(__uint_as_float(uni(__float_as_uint(sd->inc.x))), __uint_as_float(uni(__float_as_uint(sd->inc.y))));
    c.sign = __uint_as_float(uni(__float_as_uint(sd->sign)));
    c.tile = tile;
    const unsigned nb = uni(sd->n_blocks);
    c.n_out = ((unsigned)tile * T >= nb) ? 0 : (int)min((unsigned)T, nb - (unsigned)tile * T);
    const unsigned long long qf = ((unsigned long long)uni((unsigned)((unsigned long long)sd->q_first >> 32)) << 32) |
                                  uni((unsigned)(unsigned long long)sd->q_first);
    const long long qlo = (long long)qf + (long long)tile * T - 31;         // first input block of the tile
    c.first_valid = (qlo >= 0) ? 0 : ((-qlo * D > (long long)Geo::NSAMP) ? Geo::NSAMP : (int)(-qlo * D));
    c.cap = uni(sd->ring_cap);
    long long b = (long long)uni(sd->lo_mod) + ((long long)tile * T - 31) * D;   // > -cap, < 2*cap
    if (b < 0) b += c.cap;
    if (b >= (long long)c.cap) b -= c.cap;
    c.base = (unsigned)b;
    const long long c0 = (qlo >= 0) ? (qlo / kCk) : -((kCk - 1 - qlo) / kCk);     // floor(qlo/kCk)
    c.ck_first = (int)c0;
    c.pb0 = (int)(c0 * kCk - qlo);                                              // in (-kCk, 0]
    // A tile past the channel's own pending blocks (the launch's tile count comes from the channel with the MOST pending blocks): nothing
    // is computed for it, but the persistent kernels issue an item's loads before they look at n_out.  Its positions above are
    // meaningless -- `b` may lie beyond one wrap of a SMALLER ring, the checkpoint index beyond this channel's table -- so point the
    // loads at the start of the ring and of the table, which always exist (cap >= 2 NSAMP, cwslg_receiver_open; tables hold >= NCK + 4).
    if (c.n_out == 0) { c.base = 0; c.ck_first = 0; c.pb0 = 0; c.first_valid = 0; }
}

template <int D, int T, int NT>
__device__ __forceinline__ void issue_tile_loads(const TileCtx<D, T> &c, int tid, v4f (&xs)[(DemodGeom<D, T>::NSAMP + 2 * NT - 1) / (2 * NT)],
                                                 float2 &ck, v4f &tn)
{
    using Geo = DemodGeom<D, T>;
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    const CWSLG_GLOBAL v4f *ring4 = as_global(reinterpret_cast<const v4f *>(c.ring));
    if (c.base + (unsigned)Geo::NSAMP <= c.cap) {
        // the tile does not cross the end of the ring (all but one tile per ring revolution): one uniform base, lane offset
        // tid, no per-load wrap arithmetic (it was 6 VALU instructions per load)
        const CWSLG_GLOBAL v4f *p = ring4 + (c.base >> 1);
#pragma unroll
        for (int it = CWSLG_DIAG_NOHALO; it < NIT - 1; ++it) xs[it] = p[tid + it * NT];
        int r = 2 * tid + (NIT - 1) * 2 * NT;
        if (r > Geo::NSAMP - 2) r = Geo::NSAMP - 2;            // clamp: the load is unconditional
        xs[NIT - 1] = p[r >> 1];
    } else {
#pragma unroll
        for (int it = CWSLG_DIAG_NOHALO; it < NIT; ++it) {
            int r = 2 * tid + it * 2 * NT;
            if (r > Geo::NSAMP - 2) r = Geo::NSAMP - 2;
            unsigned idx = c.base + (unsigned)r;
            if (idx >= c.cap) idx -= c.cap;
            xs[it] = ring4[idx >> 1];
        }
    }
    // unconditional (clamped) checkpoint load: a predicated load would be merged with a default right away,
    // and that copy makes hipcc wait for every load issued so far
    int cidx = c.ck_first + ((tid < Geo::NCK) ? tid : 0);
    if (cidx < 0) cidx = 0;
    const v2f t = as_global(reinterpret_cast<const v2f *>(c.ckpt))[cidx];
    ck = make_float2(t.x, t.y);
    tn = as_global(reinterpret_cast<const v4f *>(c.tone))[((2 * tid) % D) >> 1];   // tone[m0], tone[m0+1]
}

template <int D, int T, int NT, int MODE>
__global__ __launch_bounds__(NT, (T <= 192 ? 5 : 4)) void demod_kernel(const ChanWork *__restrict__ works,
                                                                     const float *__restrict__ taps,
                                                                     int tiles_x, int n_ch, unsigned long long *__restrict__ clk = nullptr)
{
    using Geo = DemodGeom<D, T>;
    constexpr bool PERSIST = (MODE == 1);       // in-place prefetch of the next item
    constexpr bool LOOP = (MODE == 2);          // persistent workgroup, loads issued at the top of every item
    constexpr int G = Geo::G;
    constexpr int GL = Geo::GL;
    constexpr int PR = Geo::PR;
    constexpr int NWAVE = NT / 64;
    constexpr int NG = 64 / GL;                 // lane groups per wave
    constexpr int CPW = NG / 2;                 // 16-output chunks (x2 planes) per wave iteration
    constexpr int NIT = (Geo::NSAMP + 2 * NT - 1) / (2 * NT);
    static_assert(GL <= 16 && T % (32 * CPW) == 0 && (2 * NT) % D == 0, "geometry");
    static_assert(kDescWords <= 64 && NT >= 128, "descriptor staging uses one wave");
    static_assert(Geo::NCK <= NT, "phase 0: one phasor checkpoint per lane");

    __shared__ __attribute__((aligned(16))) float s_plane[2 * Geo::PLANE_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_aux[Geo::AUX_FLOATS];
    __shared__ __attribute__((aligned(16))) unsigned s_desc[2][kDescWords];
    float2 *s_phase = reinterpret_cast<float2 *>(s_aux);
#ifdef CWSLG_STAMP
    bool stamp_on = !PERSIST;          // persistent mode: stamp the 100th tile of each workgroup only
    int stamp_iter = 0;
#endif
    STAMP(0);

    // XCD-aware persistent schedule: XCD x owns items [x*per_xcd, (x+1)*per_xcd); its workgroups walk them with
    // stride n_slots, so neighbouring tiles (which share 31 blocks of halo) run on the same XCD at about the same time.
    const int total = (tiles_x < 0 ? -tiles_x : tiles_x) * n_ch;
    const int per_xcd = (total + 7) >> 3;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, n_slots = gridDim.x >> 3;
    const int hi_item = min((xcd + 1) * per_xcd, total);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Which pair of samples a lane loads, mixes and scatters: "mix lane" mt handles samples 2 mt + 2 NT it.  Round 4 measured where this
    // kernel's LDS bank-conflict cycles (20 % of its LDS-active cycles) come from with builds that drop one LDS phase each
    // (-DCWSLG_DIAG_LDS=1..3, scripts/gpu_r4_ldsdiag.sh): 70 % are the mix's ds_write_b64 scatter, 4 % the FIR's ds_read_b128 -- and the
    // launch takes the same 2.67 ms with the scatter's writes removed ALTOGETHER (2.671 against 2.676 ms), so they are not on the critical
    // path.  (A lane order meant to spread the sixteen lanes of a write over sixteen bank pairs, mt = tid ^ ((tid & 8) << 1), changed
    // neither the counter nor the time: the scatter's conflicts are not the 2-way pattern a 32-bank model predicts.)
    const int mt = tid;
    const int k = lane % GL;                    // this lane's branch pair (2k, 2k+1)
    int item = xcd * per_xcd + slot;
    if (item >= hi_item) return;
    // The shader clock in the middle of a timed launch (cwslg_set_timing): ONE workgroup, the one in the middle of the grid, reads s_memtime and
    // s_memrealtime when it starts and when it ends (see demod_exact3_kernel); untimed launches pass clk = nullptr and execute none of it.
    const bool clk_wg = clk != nullptr && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0;
    if (clk_wg) {
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[0] = t_;
        as_global_rw(clk)[1] = r_;
    }
    float2 tap[16];                             // H[2k][v], H[2k+1][v] = h[G*v + 2k], h[G*v + 2k + 1]
#pragma unroll
    for (int v = 0; v < 16; ++v) tap[v] = *reinterpret_cast<const float2 *>(taps + G * v + 2 * k);

    // the first descriptor is read straight from global memory (scalar loads: nothing has been stored yet);
    // later ones are staged through LDS so that no vector-memory read sits between the prefetch and its use
    const CWSLG_GLOBAL unsigned *wwords = as_global(reinterpret_cast<const unsigned *>(works));
    int cb = 0;
    TileCtx<D, T> cur;
    int ich, itile;
    item_to_ch_tile(item, tiles_x, n_ch, ich, itile);
    decode_item<D, T>(works + ich, itile, cur);
    v4f xs[NIT];
    float2 ck;
    v4f tn;
    STAMP(7);
    issue_tile_loads<D, T, NT>(cur, mt, xs, ck, tn);
    STAMP(1);

    // per-thread LDS addresses of the scatter: r = 2*tid + 2*NT*it  ->  pair-row (r % G)/2 (constant), column w0 + WSTEP*it
    const int r0 = 2 * mt;
    float *p0 = s_plane + ((r0 % G) >> 1) * PR + 2 * (r0 / G);                      // plane 0: rel = r
    const int rel1 = r0 - D + 2 * NT;                                               // plane 1: rel = r - D, taken at it = 1
    float *p1 = s_plane + Geo::PLANE_FLOATS + ((rel1 % G) >> 1) * PR + 2 * (rel1 / G - (2 * NT) / G);
    constexpr int WSTEP = (2 * NT) / G;                                             // columns per iteration

    // PERSIST: descriptors run two items ahead of the tile being computed: `dword` (one lane = one descriptor word)
    // holds item+2*n_slots' while s_desc[cb^1] holds item+n_slots', so no descriptor read is ever waited for.
    unsigned dword = 0;
    const bool desc_lane = PERSIST && tid >= 64 && tid < 64 + kDescWords;
    if (PERSIST) {
        const int i1 = item + n_slots;
        if (desc_lane && i1 < hi_item) { item_to_ch_tile(i1, tiles_x, n_ch, ich, itile); s_desc[1][tid - 64] = wwords[(size_t)ich * kDescWords + (tid - 64)]; }
        const int i2 = item + 2 * n_slots;
        if (desc_lane && i2 < hi_item) { item_to_ch_tile(i2, tiles_x, n_ch, ich, itile); dword = wwords[(size_t)ich * kDescWords + (tid - 64)]; }
    }

    for (;;) {
        const int nitem = item + n_slots;
        const bool has_next = PERSIST && nitem < hi_item;                           // wave-uniform
        // ---- phase 0: bit-exact phasor for the tile's T+31 blocks (lanes 0..NCK-1, <=kCk un-fused steps each)
        {
            const int cidx = cur.ck_first + mt;
            if (mt < Geo::NCK && cidx >= 0) {
                float2 p = ck;
                const int pbase = cur.pb0 + kCk * mt;
#pragma unroll
                for (int s = 0; s < kCk; ++s) {
                    const int pb = pbase + s;
                    if (pb >= 0 && pb < Geo::NBLK) s_phase[pb] = p;
                    p = cmul_exact(p, cur.inc);
                }
            }
        }
        lds_barrier();                                // s_phase ready; s_desc[cb^1] (written an iteration ago) visible
        STAMP(2);
#ifdef CWSLG_STAMP
        if (!PERSIST) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        STAMP(3);
#endif

        // the descriptor two items ahead, loaded an iteration ago, moves into the slot the current item vacated (cur's
        // fields live in registers); then fetch the one after it.  Placed BEFORE this iteration's prefetch loads so that
        // the wait for `dword` cannot drain them (vmcnt retires in order).
        if (PERSIST && desc_lane) {
            if (nitem + n_slots < hi_item) s_desc[cb][tid - 64] = dword;
            const int i3 = item + 3 * n_slots;
            if (i3 < hi_item) { int c3, t3; item_to_ch_tile(i3, tiles_x, n_ch, c3, t3); dword = wwords[(size_t)c3 * kDescWords + (tid - 64)]; }
        }
        // ---- the NEXT item: decode now, its loads are issued in place as phase 1 frees the registers
        TileCtx<D, T> nxt = cur;
        const CWSLG_GLOBAL v4f *nring4 = nullptr;
        if (has_next) {
            int nch, ntile;
            item_to_ch_tile(nitem, tiles_x, n_ch, nch, ntile);
            decode_item<D, T>(reinterpret_cast<const ChanWork *>(s_desc[cb ^ 1]), ntile, nxt);
            nring4 = as_global(reinterpret_cast<const v4f *>(nxt.ring));
            int cidx = nxt.ck_first + ((mt < Geo::NCK) ? mt : 0);
            if (cidx < 0) cidx = 0;
            const v2f t = as_global(reinterpret_cast<const v2f *>(nxt.ckpt))[cidx];       // ck was consumed by phase 0
            ck = make_float2(t.x, t.y);
        }

        if (cur.n_out > 0 || has_next) {
        // ---- phase 1: mix (x*tone)*phase, scatter Re -> plane 0, Im -> plane 1
        {
            float2 ph[NIT];
#pragma unroll
            for (int it = CWSLG_DIAG_NOHALO; it < NIT; ++it) {
                int blk = (2 * mt) / D + it * (2 * NT / D);
                if (blk > Geo::NBLK - 1) blk = Geo::NBLK - 1;
                ph[it] = s_phase[blk];
            }
            const float2 tn0 = make_float2(tn.x, tn.y), tn1 = make_float2(tn.z, tn.w);
            if (has_next) tn = as_global(reinterpret_cast<const v4f *>(nxt.tone))[((2 * mt) % D) >> 1];
            const int fv = cur.first_valid;
            // two copies of the loop behind ONE scalar branch: with the origin test inside a single loop hipcc if-converts it, and
            // its four moves and the compare are then issued (under an empty exec mask) for every load of every tile
            auto mix = [&](auto slow_tag) {
                constexpr bool SLOW = decltype(slow_tag)::value;
#pragma unroll
                for (int it = CWSLG_DIAG_NOHALO; it < NIT; ++it) {
                    const int r = 2 * mt + it * 2 * NT;
                    const v4f x = xs[it];
                    if (has_next) {                                  // in-place prefetch: xs[it] is free from here on
                        int rn = (r > Geo::NSAMP - 2) ? Geo::NSAMP - 2 : r;
                        unsigned idx = nxt.base + (unsigned)rn;
                        if (idx >= nxt.cap) idx -= nxt.cap;
                        xs[it] = nring4[idx >> 1];
                    }
                    float ar = __builtin_fmaf(x.x, tn0.x, -(x.y * tn0.y));
                    float ai = __builtin_fmaf(x.x, tn0.y, x.y * tn0.x);
                    float y0r = __builtin_fmaf(ar, ph[it].x, -(ai * ph[it].y));
                    float y0i = __builtin_fmaf(ar, ph[it].y, ai * ph[it].x);
                    ar = __builtin_fmaf(x.z, tn1.x, -(x.w * tn1.y));
                    ai = __builtin_fmaf(x.z, tn1.y, x.w * tn1.x);
                    float y1r = __builtin_fmaf(ar, ph[it].x, -(ai * ph[it].y));
                    float y1i = __builtin_fmaf(ar, ph[it].y, ai * ph[it].x);
                    if (SLOW) {                                      // only the first tiles of a slot
                        if (r < fv) { y0r = 0.f; y0i = 0.f; y1r = 0.f; y1i = 0.f; }
                    }
                    const bool in0 = (it < NIT - 1) || (r < G * (T / 2 + 15));          // plane 0 drops the last D samples
                    const bool in1 = (it > 0) ? ((it < NIT - 1) || (r < Geo::NSAMP)) : (r >= D);   // plane 1 drops the first D
#if CWSLG_DIAG_LDS != 1          // (diagnostic build 1: the mix without its LDS writes -- which phase owns the bank-conflict cycles)
                    if (in0) *reinterpret_cast<float2 *>(p0 + 2 * it * WSTEP) = make_float2(y0r, y1r);
                    if (in1) *reinterpret_cast<float2 *>(p1 + 2 * it * WSTEP) = make_float2(y0i, y1i);
#else
                    if (y0r + y1r + y0i + y1i == 123.456f && in0 && in1) p0[0] = y0r;      // keep the arithmetic alive
#endif
                }
            };
            if (PERSIST) mix(std::true_type{});          // (one copy only: with two, the in-place prefetch spills at 128 VGPRs)
            else if (fv != 0) mix(std::true_type{});
            else mix(std::false_type{});
        }
        lds_barrier();
        STAMP(4);

        // ---- phase 2: branch-pair FIRs + cross-lane reduction.  s_aux becomes the output row.
        if (cur.n_out > 0) {
            const int g = lane / GL;
            const int pl = g & 1;
            const float sgn_plane = pl ? -cur.sign : 1.0f;
            for (int it = wv; it < T / (32 * CPW); it += NWAVE) {
                const int chunk = it * CPW + (g >> 1);
                const float4 *src = reinterpret_cast<const float4 *>(
                    s_plane + pl * Geo::PLANE_FLOATS + k * PR + 2 * 16 * chunk);
                float acc[16];
#if CWSLG_FIR_PK
                // The lane's two branches accumulate side by side in ONE register pair per output: (x_even, x_odd) is an aligned
                // pair of the float4 read from LDS, (h_even, h_odd) an aligned pair of tap registers, so the two FMAs are one
                // v_pk_fma_f32: 256 instead of 512 FMA instructions per wave and tile, same VGPR count.  Measured (same box, 512 / 4096
                // slots): 2.558-2.565 against 2.563-2.573 ms, 20.51-20.63 against 20.64-20.68 ms per launch -- within noise.  Neither
                // the issue slots nor (section 10 of DESIGN.md) the instruction count bound this kernel; the FLOPs and bytes, which
                // set its power, do.  Off by default because it changes the summation order (evens + odds) for no gain.
                v2f acc2[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[r] = v2f{0.0f, 0.0f};
                const v4f *src4 = reinterpret_cast<const v4f *>(src);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const v4f c4 = src4[q];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = 2 * q + h;
                        const v2f x2 = h ? v2f{c4.z, c4.w} : v2f{c4.x, c4.y};
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const int r = j - v;
                            if (r >= 0 && r < 16) acc2[r] = __builtin_elementwise_fma(x2, v2f{tap[v].x, tap[v].y}, acc2[r]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = acc2[r].x + acc2[r].y;
#else
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                // column j = w - 16*chunk (0..30) feeds acc[j - v], v = 0..15; one float4 = columns 2q, 2q+1 x both branches
#pragma unroll
                for (int q = 0; q < 16; ++q) {
#if CWSLG_DIAG_LDS != 2          // (diagnostic build 2: the FIR without its LDS reads)
                    const float4 c4 = src[q];
#else
                    const float4 c4 = make_float4(tap[q].x, tap[q].y, tap[15 - q].x, tap[15 - q].y);
#endif
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int j = 2 * q + h;
                        const float xe = h ? c4.z : c4.x, xo = h ? c4.w : c4.y;
#pragma unroll
                        for (int v = 0; v < 16; ++v) {
                            const int r = j - v;
                            if (r >= 0 && r < 16) {
                                acc[r] = __builtin_fmaf(xe, tap[v].x, acc[r]);
                                acc[r] = __builtin_fmaf(xo, tap[v].y, acc[r]);
                            }
                        }
                    }
                }
#endif
                const int rbase = reduce_lanes<GL>(acc, k);
                constexpr int NV = 16 / GL;
#pragma unroll
                for (int r = 0; r < NV; ++r) {
                    const int wq = 16 * chunk + rbase + r;
                    const float s = (wq & 1) ? -sgn_plane : sgn_plane;
#if CWSLG_DIAG_LDS != 3          // (diagnostic build 3: no output staging through LDS)
                    s_aux[2 * wq + pl] = s * acc[r];
#else
                    if (s * acc[r] == 123.456f) s_aux[0] = 1.0f;
#endif
                }
            }
        }
        lds_barrier();
        STAMP(5);

        // ---- epilogue: whole-row store + frame peak
        if (cur.n_out > 0) {
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
        STAMP(6);
        }   // work in this iteration
        if (LOOP) {
            item = nitem;
            if (item >= hi_item) break;
            lds_barrier();                                       // s_aux (= s_phase) is rewritten by the next phase 0
            item_to_ch_tile(item, tiles_x, n_ch, ich, itile);
            decode_item<D, T>(works + ich, itile, cur);
            issue_tile_loads<D, T, NT>(cur, mt, xs, ck, tn);
            continue;
        }
        if (!has_next) break;
        lds_barrier();                                           // s_aux (= s_phase) is rewritten by the next phase 0
        cur = nxt;
        item = nitem;
        cb ^= 1;
#ifdef CWSLG_STAMP
        ++stamp_iter;
        stamp_on = (stamp_iter == 100);
        STAMP(0); STAMP(7); STAMP(1);
#endif
    }
    if (clk_wg) {
        unsigned long long t_, r_;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[2] = t_;
        as_global_rw(clk)[3] = r_;
    }
}

// (Round 5: the retired demod kernels -- demod_mfma1p_kernel, demod_mfma_bf16_kernel, ring_probe_kernel, demod_exact_kernel, demod_exact2_kernel --
// live in lab/demod_lab_kernels.hpp, which only the lab build includes.)

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

// ---------------------------------------------------------------------------------------------
// demod_exact5_kernel<D> (192 / 96 / 48 kHz: D = 16 / 8 / 4 samples per block; round 5): the reference's arithmetic with LANE = STREAM -- a wave serves 32 consecutive segments of seg_len
// outputs of ONE channel, time runs along the lane, and everything a stream needs between two tiles (its 16 samples, the 32 x 2 running sums,
// the workspace chain, its mixer phase) stays in registers.  The un-fused products fl(y * h) (SSBD.hpp:167-168) of a block against all 32 tap
// blocks are ONE K = 1 matrix instruction per sample (v_mfma_f32_32x32x1_2b_f32, C = 0: bit-identical to v_mul_f32, scripts/micro/mfma_k1.hip);
// the ordered sums, sum * phase and the workspace accumulation are plain v_add_f32 / v_mul_f32 in the reference's order.  Why this shape, what
// the matrix instruction does and does not buy (it shares the FP32 lanes with the VALU: no second pipe), the register map and the zero-sign
// argument: scripts/gen_exact5_asm.py and DESIGN.md 4.1d; profiles/r5_mfma_k1.txt.  The wave's whole life is one generated assembly statement
// (exact5_asm.inc), executed on the CPU against the oracle before it ever reached a GPU (tests/test_exact5_stream.py, tests/wave_emulator.py).
//   Work item = (channel, chunk of 32 x seg_len outputs); one wave per item, four waves per workgroup, no barrier anywhere (a wave's LDS
//   rows are its own: the 32 x 128 bytes of a tile are loaded coalesced -- eight lanes per 128-byte line -- and read back lane = stream).
//   A stream starts 32 blocks before its first output (the workspace needs an output's 32 blocks); the host therefore hands this kernel only
//   work whose 32-block history exists: q_first >= 32 (the first 32 outputs of a fresh demodulator go through demod_exact4_kernel).
//   Requires q_first, n_blocks, seg_len multiples of 4, lo_mod and the ring's length multiples of 4 D samples (the push granularity) and the ring
//   shorter than 4 GiB (host-checked).  At 96 / 48 kHz a block is 64 / 32 bytes: the wave still moves 128-byte rows (two / four tiles per row).
#include "exact5_asm.inc"
constexpr int kExact5Waves = 4;
template <int D>
__global__ __launch_bounds__(64 * kExact5Waves, 2) void demod_exact5_kernel(const ChanWork *__restrict__ works, const float *__restrict__ taps, int chunks_x,
                                                                              int n_ch, int seg_len, unsigned long long *__restrict__ clk)
{
    static_assert(D == 16 || D == 8 || D == 4, "192 / 96 / 48 kHz");
    constexpr int TILES = D == 16 ? EXACT5_D16_TILES_PER_ITER : D == 8 ? EXACT5_D8_TILES_PER_ITER : EXACT5_D4_TILES_PER_ITER;
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[kExact5Waves * 2 * EXACT5_ASM_BUF_BYTES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = (int)blockIdx.x * kExact5Waves + wv;
    const int n_items = (chunks_x < 0 ? -chunks_x : chunks_x) * n_ch;
    if (item >= n_items) return;
    int ch, chunk;
    item_to_ch_tile(item, chunks_x, n_ch, ch, chunk);
    ch = (int)uni((unsigned)ch); chunk = (int)uni((unsigned)chunk);
    const CWSLG_CONST ChanWork *cw = as_const(works + ch);
    const unsigned n_blocks = cw->n_blocks, cap = cw->ring_cap, lo_mod = cw->lo_mod;
    const long long q_first = cw->q_first;
    const long long first_seg = (long long)chunk * 32;
    if (first_seg * seg_len >= (long long)n_blocks) return;                           // a chunk past this channel's pending blocks
    const unsigned most = (unsigned)min((long long)seg_len, (long long)n_blocks - first_seg * seg_len);   // outputs of the wave's first (fullest) stream
    // ring byte offset of the first sample of stream s (its 32-block warm-up included): lo_mod + 16 (first output - 32), modulo the ring
    auto stream_pos = [&](int s) -> unsigned {
        long long rel = ((first_seg + s) * (long long)seg_len - 32) * D + (long long)lo_mod;
        rel %= (long long)cap;
        if (rel < 0) rel += cap;
        return (unsigned)rel * 8u;
    };
    const int j = lane & 31;
    unsigned off0 = stream_pos(0 + (lane >> 3)) + (unsigned)(lane & 7) * 16u, off1 = stream_pos(8 + (lane >> 3)) + (unsigned)(lane & 7) * 16u;
    unsigned off2 = stream_pos(16 + (lane >> 3)) + (unsigned)(lane & 7) * 16u, off3 = stream_pos(24 + (lane >> 3)) + (unsigned)(lane & 7) * 16u;
    const unsigned pc16 = (unsigned)(lane & 7) * 16u, capl = cap * 8u + pc16;
    const unsigned lds0 = (unsigned)(size_t)(s_rows + wv * 2 * EXACT5_ASM_BUF_BYTES);
    const unsigned ldsr = lds0 + (unsigned)j * EXACT5_ASM_ROW_BYTES, ldsw = lds0 + (unsigned)(lane >> 3) * EXACT5_ASM_ROW_BYTES + pc16;
    const long long my_first = (first_seg + j) * (long long)seg_len;
    int rem = (int)max(0ll, min((long long)seg_len, (long long)n_blocks - my_first));
    const long long q_start = q_first + my_first - 32;
    const unsigned ckoff = rem > 0 ? (unsigned)(q_start / kCk) * 8u : 0u;                // an idle stream: any entry of the table
    unsigned outoff = (unsigned)my_first * 4u;
    // the matrix instruction's row i (operand lane i & 31) lands in the half (i >> 2) & 1, register (i & 3) + 4 (i >> 3): give row i tap block
    // n = register + 16 half, so that a lane of the lower half holds tap blocks 0..15 in register order and the upper half 16..31
    const unsigned tapoff = (unsigned)((j & 3) + 4 * (j >> 3) + 16 * ((j >> 2) & 1)) * (unsigned)(4 * D);
    const float sign = __uint_as_float(uni(__float_as_uint(cw->sign)));
    const float nsign = -sign;
    const float incre = __uint_as_float(uni(__float_as_uint(cw->inc.x))), incim = __uint_as_float(uni(__float_as_uint(cw->inc.y)));
    const float2 *ring = uni_ptr(cw->ring), *ckpt = uni_ptr(cw->ckpt), *tone = uni_ptr(cw->tone);
    float *out = uni_ptr(cw->out);
    unsigned *peak_word = uni_ptr(cw->peak);
    const float *taps_u = uni_ptr(taps);
    int warm = EXACT5_ASM_WARM_STORES, iters = (int)((32u + most + TILES - 1) / TILES);       // 32 warm-up tiles, then one tile per output
    const unsigned long long hmask = 0xFFFFFFFF00000000ull;
    unsigned long long esave;
    float peak;
    const bool stamp = clk != nullptr && blockIdx.x == 0 && wv == 0;
    if (stamp && lane == 0) {
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[0] = t_;
        as_global_rw(clk)[1] = r_;
    }
#define EXACT5_STATEMENT(TEXT)                                                                                                                             \
    asm volatile(TEXT                                                                                                                                      \
                 : [off0] "+v"(off0), [off1] "+v"(off1), [off2] "+v"(off2), [off3] "+v"(off3), [rem] "+v"(rem), [outoff] "+v"(outoff), [peak] "=&v"(peak), \
                   [esave] "=&s"(esave), [warm] "+s"(warm), [iters] "+s"(iters)                                                                            \
                 : [capl] "v"(capl), [pc16] "v"(pc16), [ldsr] "v"(ldsr), [ldsw] "v"(ldsw), [ckoff] "v"(ckoff), [tapoff] "v"(tapoff),                       \
                   [ring] "s"(ring), [taps] "s"(taps_u), [tone] "s"(tone), [ckpt] "s"(ckpt), [out] "s"(out),                                               \
                   [incre] "s"(incre), [incim] "s"(incim), [sign] "s"(sign), [nsign] "s"(nsign), [hmask] "s"(hmask)                                        \
                 : EXACT5_ASM_CLOBBERS)
    if constexpr (D == 16) EXACT5_STATEMENT(EXACT5_D16_PROLOGUE_ASM EXACT5_D16_LOOP_ASM EXACT5_D16_EPILOGUE_ASM);
    else if constexpr (D == 8) EXACT5_STATEMENT(EXACT5_D8_PROLOGUE_ASM EXACT5_D8_LOOP_ASM EXACT5_D8_EPILOGUE_ASM);
    else EXACT5_STATEMENT(EXACT5_D4_PROLOGUE_ASM EXACT5_D4_LOOP_ASM EXACT5_D4_EPILOGUE_ASM);
#undef EXACT5_STATEMENT
    const float mx = wave_max_dpp(peak);
    if (lane == 0) publish_peak(peak_word, mx);
    if (stamp && lane == 0) {
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[2] = t_;
        as_global_rw(clk)[3] = r_;
    }
}

// ---------------------------------------------------------------------------------------------
// demod_transition_kernel: the (at most 32) outputs after a phase-continuous retune, in ProcessBlock's own order (SSBD.hpp:160-183)
// whatever the context's mode -- they are a handful per retune.  grid = works, 64 threads: thread = output.
template <int D>
__global__ __launch_bounds__(64) void demod_transition_kernel(const TransWork *__restrict__ works, const float *__restrict__ taps)
{
    const TransWork *w = works + blockIdx.x;
    const int o = threadIdx.x;
    if (o >= w->n_out) return;
    const int b = w->o_first + o;                              // output block, relative to b0
    const CWSLG_GLOBAL float2 *ring = as_global(w->ring);
    float wr = 0.0f, wi = 0.0f;                                // the workspace slot, zero after its last read-out (:178)
    for (int n = 0; n < 32; ++n) {                             // oldest block first, as the blocks arrived
        const int beta = b - 31 + n;                           // block, relative to b0
        if (beta < -w->blocks_before) continue;                // before the demodulator existed: nothing was ever added
        const bool old = beta < 0;
        const float2 *tone = old ? w->tone_old : w->tone_new;
        const float2 ph = old ? w->phase_old[beta + 32] : w->phase_new[beta];
        long long pos = (long long)w->pos_b0 + (long long)beta * D;
        if (pos < 0) pos += w->ring_cap;
        if (pos >= (long long)w->ring_cap) pos -= w->ring_cap;
        float sr = 0.0f, si = 0.0f;
        for (int m = 0; m < D; ++m) {                          // :166-169, un-fused
            unsigned idx = (unsigned)pos + (unsigned)m;
            if (idx >= w->ring_cap) idx -= w->ring_cap;
            const float2 x = make_float2(ring[idx].x, ring[idx].y);
            const float2 t = cmul_exact(x, tone[m]);
            const float h = taps[m + D * n];
            sr = sr + t.x * h;
            si = si + t.y * h;
        }
        const float2 pr = cmul_exact(make_float2(sr, si), ph);  // sum * phase (:170)
        wr = wr + pr.x;
        wi = wi + pr.y;
    }
    // Iterate (:131-134): block position mod 4 (b0 is a multiple of 4 blocks from the origin)
    float v;
    switch (b & 3) {
    case 0: v = wr; break;
    case 1: v = -wi * w->sign; break;
    case 2: v = -wr; break;
    default: v = wi * w->sign; break;
    }
    as_global_rw(w->out)[o] = v;
    publish_peak(w->peak, fabsf(v));
}

// ---------------------------------------------------------------------------------------------
// Slot finalise: prepareAudio + float->int16 (Instance.cpp:294-338, 238-241), bit-exact:
//   factor = 32767.0f / (peak + 1.0f); factor *= scale;  buf[k] *= factor;  (int16)(buf[k] + 0.5f)
// The peak is max|audio| (max(maxVal, |minVal|) == max|x| for every frame).
// Samples at and beyond n_valid are the reference's zero tail.
constexpr int kFinChunks = 4;          // 8-sample chunks per thread of finalize_kernel: four 32-byte reads in flight per lane
template <int NT>
__global__ __launch_bounds__(NT) void finalize_kernel(const FinWork *__restrict__ works)
{
    // the descriptor's pointers are HBM addresses: say so (global_load / global_store instead of flat_*, which also count in lgkmcnt)
    const FinWork *fw = works + blockIdx.y;
    const CWSLG_GLOBAL float *frame = as_global(fw->frame);
    CWSLG_GLOBAL int16_t *out = as_global_rw(fw->out);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (fw->peak_next) *as_global_rw(fw->peak_next) = 0u;
    }
    // a workgroup covers kFinChunks x NT x 8 consecutive samples; chunk c of a thread is NT x 8 samples behind chunk c - 1 (every wave-level
    // access is one contiguous 2 KB / 1 KB run).  Round 3: one chunk per thread was 360 000 workgroups of two loads and a store each
    // per 4096-slot boundary (0.85 ms for 4.4 GB); four chunks per thread, all loads issued first: see DESIGN.md 4.2.
    const unsigned base = blockIdx.x * (unsigned)(kFinChunks * NT * 8) + threadIdx.x * 8u;
    if (!fw->emit || blockIdx.x * (unsigned)(kFinChunks * NT * 8) >= fw->frame_len) return;
    const unsigned nv = fw->n_valid, flen = fw->frame_len;
    v4f a[kFinChunks], b[kFinChunks];
#pragma unroll
    for (int c = 0; c < kFinChunks; ++c) {
        const unsigned i0 = base + (unsigned)c * (NT * 8);
        if (i0 + 8 <= nv) {
            a[c] = *reinterpret_cast<const CWSLG_GLOBAL v4f *>(frame + i0);
            b[c] = *reinterpret_cast<const CWSLG_GLOBAL v4f *>(frame + i0 + 4);
        } else {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (i0 + k < nv) ? frame[i0 + k] : 0.0f;
            a[c] = v4f{v[0], v[1], v[2], v[3]};
            b[c] = v4f{v[4], v[5], v[6], v[7]};
        }
    }
    const float peak = __uint_as_float(*as_global(fw->peak));
    float factor = 32767.0f / (peak + 1.0f);
    factor = factor * fw->scale;
    if (base == 0 && fw->factor_out) *as_global_rw(fw->factor_out) = factor;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int c = 0; c < kFinChunks; ++c) {
        const unsigned i0 = base + (unsigned)c * (NT * 8);
        if (i0 >= flen) break;
        const float v[8] = {a[c].x, a[c].y, a[c].z, a[c].w, b[c].x, b[c].y, b[c].z, b[c].w};
        int q[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float scaled = v[k] * factor;                 // buf[k] *= factor
            const float biased = scaled + 0.5f;                 // + 0.5f
            q[k] = (int)biased;                                 // C truncation toward zero, then narrowed to int16
        }
        v4u pk;
        pk.x = ((unsigned)q[0] & 0xFFFFu) | ((unsigned)q[1] << 16);
        pk.y = ((unsigned)q[2] & 0xFFFFu) | ((unsigned)q[3] << 16);
        pk.z = ((unsigned)q[4] & 0xFFFFu) | ((unsigned)q[5] << 16);
        pk.w = ((unsigned)q[6] & 0xFFFFu) | ((unsigned)q[7] << 16);
        const unsigned rem = flen - i0;
        if (rem >= 8) {
            *reinterpret_cast<CWSLG_GLOBAL v4u *>(out + i0) = pk;
        } else {
            for (unsigned k = 0; k < rem; ++k) out[i0 + k] = (int16_t)q[k];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Portable synthetic IQ, bit-identical to oracle/cwsl_oracle.c:orc_synth_* (integer Irwin-Hall
// noise, exact in float; tones by uint32 phase accumulator into a host-made 4096-entry table).
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float noise_from_bits(unsigned long long r)
{
    const int s = (int)(r & 0xFFFF) + (int)((r >> 16) & 0xFFFF) + (int)((r >> 32) & 0xFFFF) +
                  (int)((r >> 48) & 0xFFFF) - 131070;
    return (float)s * 0.03125f;
}

struct SynthArgs {
    float2 *ring;
    unsigned ring_cap;
    unsigned long long ring_pos;      // ring index of the first generated sample
    unsigned long long first_sample;  // stream index of the first generated sample
    unsigned long long seed;
    unsigned n;
    int n_tones;
    float amp;
    unsigned step[8];                 // cycles/sample * 2^32
};

__global__ void synth_kernel(SynthArgs a, const float2 *__restrict__ sincos_tab)
{
    const unsigned k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n) return;
    const unsigned long long i = a.first_sample + k;
    float re = noise_from_bits(mix64(a.seed ^ (2ull * i) * 0xD1342543DE82EF95ull));
    float im = noise_from_bits(mix64(a.seed ^ (2ull * i + 1ull) * 0xD1342543DE82EF95ull));
    for (int t = 0; t < a.n_tones; ++t) {
        const unsigned ph = (unsigned)(i * (unsigned long long)a.step[t]);
        const float2 cs = sincos_tab[ph >> 20];
        re = re + a.amp * cs.x;       // unfused, like the oracle
        im = im + a.amp * cs.y;
    }
    unsigned long long p = a.ring_pos + k;
    if (p >= a.ring_cap) p -= a.ring_cap;
    a.ring[p] = make_float2(re, im);
}

} // namespace cwslg

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
    unsigned     tail_end;    // (the fused finalise of symbol_spectra_v2_kernel) samples at and beyond this index are zero in `out` ALREADY and in this
                              // frame: max(n_valid of this slot, n_valid of the slot `out` held before), so the tail beyond the last window is
                              // rewritten only where one of the two put samples there; frame_len when unknown
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
// of the six scalar operations; used where what bounds the code is what one wave can issue (the mix of the lab library's demod_exact3_kernel).
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

// Cache policy of the IQ ring's loads in demod_kernel (round 5 A/B, scripts/gpu_r5_fast_nt.sh): a private stream is read once (31 of every 287 blocks twice:
// the tile halo, by the neighbouring workgroup), so non-temporal is a candidate; demod_exact5_kernel gained 2.7 % from it.
#ifndef CWSLG_RING_NT
#define CWSLG_RING_NT 0
#endif
__device__ __forceinline__ v4f ring_ld(const CWSLG_GLOBAL v4f *p)
{
    return CWSLG_RING_NT ? __builtin_nontemporal_load(p) : *p;
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
        for (int it = CWSLG_DIAG_NOHALO; it < NIT - 1; ++it) xs[it] = ring_ld(p + tid + it * NT);
        int r = 2 * tid + (NIT - 1) * 2 * NT;
        if (r > Geo::NSAMP - 2) r = Geo::NSAMP - 2;            // clamp: the load is unconditional
        xs[NIT - 1] = ring_ld(p + (r >> 1));
    } else {
#pragma unroll
        for (int it = CWSLG_DIAG_NOHALO; it < NIT; ++it) {
            int r = 2 * tid + it * 2 * NT;
            if (r > Geo::NSAMP - 2) r = Geo::NSAMP - 2;
            unsigned idx = c.base + (unsigned)r;
            if (idx >= c.cap) idx -= c.cap;
            xs[it] = ring_ld(ring4 + (idx >> 1));
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
    if (item >= hi_item) {
        // the launch's clock slot must not stay open when its stamping workgroup has no work (drain_spans waits for end stamps while spans are
        // queued: a real-time host always has some): a sentinel pair -- non-zero end, zero start -- that the host reads as "nothing to fold in"
        if (clk != nullptr && blockIdx.x == (gridDim.x >> 1) && threadIdx.x == 0) { as_global_rw(clk)[2] = 1ull; as_global_rw(clk)[3] = 1ull; }
        return;
    }
    // The shader clock in the middle of a timed launch (cwslg_set_timing): ONE workgroup, the one in the middle of the grid, reads s_memtime and
    // s_memrealtime when it starts and when it ends (see demod_exact5_kernel); untimed launches pass clk = nullptr and execute none of it.
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

// (Round 5: demod_exact3_kernel and demod_exact4_kernel -- the tile-shaped exact kernels of rounds 3 and 4 -- live in lab/demod_lab_kernels.hpp with their
// generated FIR streams; the product's exact mode is demod_exact5_kernel<D> at every rate, the first outputs of a demodulator included.)

// ---------------------------------------------------------------------------------------------
// demod_exact5_kernel<D> (192 / 96 / 48 kHz: D = 16 / 8 / 4 samples per block; round 5): the reference's arithmetic with LANE = STREAM -- a wave serves 32 consecutive segments of seg_len
// outputs of ONE channel, time runs along the lane, and everything a stream needs between two tiles (its 16 samples, the 32 x 2 running sums,
// the workspace chain, its mixer phase) stays in registers.  The un-fused products fl(y * h) (SSBD.hpp:167-168) of a block against all 32 tap
// blocks are ONE K = 1 matrix instruction per sample (v_mfma_f32_32x32x1_2b_f32, C = 0: bit-identical to v_mul_f32, scripts/micro/mfma_k1.hip);
// the ordered sums (as v_pk_add_f32 register pairs, each half rounded on its own), sum * phase and the workspace accumulation are plain FP32 adds / products in the reference's order.  Why this shape, what
// the matrix instruction does and does not buy (it shares the FP32 lanes with the VALU: no second pipe), the register map and the zero-sign
// argument: scripts/gen_exact5_asm.py and DESIGN.md 4.1d; profiles/r5_mfma_k1.txt.  The wave's whole life is one generated assembly statement
// (exact5_asm.inc), executed on the CPU against the oracle before it ever reached a GPU (tests/test_exact5_stream.py, tests/wave_emulator.py).
//   Work item = (channel, chunk of 32 x seg_len outputs); one wave per item, four waves per workgroup, no barrier anywhere (a wave's LDS
//   rows are its own: the 32 x 128 bytes of a tile travel coalesced -- eight lanes per 128-byte line -- from the ring straight into LDS (LDS-DMA,
//   four 4 KB buffers per wave, swizzled on the source side) and are read back lane = stream).
//   A stream starts 32 blocks before its first output (the workspace needs an output's 32 blocks); where that reaches before the demodulator's
//   origin (the first 32 outputs after its creation) the missing blocks are zeros and the phase is held at (1, 0): see `hold` below.
//   Requires q_first, n_blocks, seg_len multiples of 4, lo_mod and the ring's length multiples of 4 D samples (the push granularity) and the ring
//   shorter than 4 GiB (host-checked).  At 96 / 48 kHz a block is 64 / 32 bytes: the wave still moves 128-byte rows (two / four tiles per row).
#include "exact5_asm.inc"
#ifndef CWSLG_EXACT5_WAVES
#define CWSLG_EXACT5_WAVES 4
#endif
constexpr int kExact5Waves = CWSLG_EXACT5_WAVES;       // waves per workgroup (no barrier anywhere: any count works; A/B in profiles/r5_experiments.txt)
static_assert(kExact5Waves * EXACT5_ASM_NBUF * EXACT5_ASM_BUF_BYTES <= 65536, "a workgroup's row buffers (waves x buffers x 4 KB) exceed the 64 KB static LDS limit");
template <int D>
__global__ __launch_bounds__(64 * kExact5Waves, 2) void demod_exact5_kernel(const ChanWork *__restrict__ works, const float *__restrict__ taps, int chunks_x,
                                                                              int n_ch, int seg_len, unsigned long long *__restrict__ clk)
{
    static_assert(D == 16 || D == 8 || D == 4, "192 / 96 / 48 kHz");
    constexpr int TILES = D == 16 ? EXACT5_D16_TILES_PER_ITER : D == 8 ? EXACT5_D8_TILES_PER_ITER : EXACT5_D4_TILES_PER_ITER;
    __shared__ __attribute__((aligned(16))) unsigned char s_rows[kExact5Waves * EXACT5_ASM_NBUF * EXACT5_ASM_BUF_BYTES];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = (int)blockIdx.x * kExact5Waves + wv;
    const int n_items = (chunks_x < 0 ? -chunks_x : chunks_x) * n_ch;
    // (the clock of the launch's MIDDLE workgroup: the first ones run in the boost the package allows a launch's first millisecond or two)
    const bool stamp = clk != nullptr && blockIdx.x == (gridDim.x >> 1) && wv == 0;
    // With ragged n_blocks per channel the stamping wave can have no work: it then closes the launch's clock slot with a sentinel (non-zero end
    // stamps, zero start stamps: drain_spans skips it) instead of leaving it open, which would hold every later slot back (statistics only).
    auto close_clock_slot = [&]() { if (stamp && lane == 0) { as_global_rw(clk)[2] = 1ull; as_global_rw(clk)[3] = 1ull; } };
    if (item >= n_items) { close_clock_slot(); return; }
    int ch, chunk;
    item_to_ch_tile(item, chunks_x, n_ch, ch, chunk);
    ch = (int)uni((unsigned)ch); chunk = (int)uni((unsigned)chunk);
    const CWSLG_CONST ChanWork *cw = as_const(works + ch);
    const unsigned n_blocks = cw->n_blocks, cap = cw->ring_cap, lo_mod = cw->lo_mod;
    const long long q_first = cw->q_first;
    const long long first_seg = (long long)chunk * 32;
    if (first_seg * seg_len >= (long long)n_blocks) { close_clock_slot(); return; }   // a chunk past this channel's pending blocks
    const unsigned most = (unsigned)min((long long)seg_len, (long long)n_blocks - first_seg * seg_len);   // outputs of the wave's first (fullest) stream
    // ring byte offset of the first sample of stream s (its 32-block warm-up included): lo_mod + 16 (first output - 32), modulo the ring
    auto stream_pos = [&](int s) -> unsigned {
        long long rel = ((first_seg + s) * (long long)seg_len - 32) * D + (long long)lo_mod;
        rel %= (long long)cap;
        if (rel < 0) rel += cap;
        return (unsigned)rel * 8u;
    };
    const int j = lane & 31;
    const unsigned lds0 = (unsigned)(size_t)(s_rows + wv * EXACT5_ASM_NBUF * EXACT5_ASM_BUF_BYTES);
#if EXACT5_ASM_DMA
    // LDS-DMA form (lab builds: X5_DMA=1 at generation): a load's 64 lanes land lane-linear (row r = lanes 8 r .. 8 r + 7), so the pieces of a row are
    // swizzled on the SOURCE side -- lane l of load i fetches piece (l & 7) ^ f(row), f(row) = (row >> 1) & 7 -- and a reader undoes it with its address
    const unsigned pc160 = (unsigned)((lane & 7) ^ ((lane >> 4) & 7)) * 16u, pc161 = (unsigned)((lane & 7) ^ ((4 + (lane >> 4)) & 7)) * 16u;
    unsigned off0 = stream_pos(0 + (lane >> 3)) + pc160, off1 = stream_pos(8 + (lane >> 3)) + pc161;
    unsigned off2 = stream_pos(16 + (lane >> 3)) + pc160, off3 = stream_pos(24 + (lane >> 3)) + pc161;
    const unsigned capl0 = cap * 8u + pc160, capl1 = cap * 8u + pc161;
    const unsigned ldsr = lds0 + (unsigned)j * 128u, ldsw = lds0 + (unsigned)lane * 16u, fj16 = (unsigned)((j >> 1) & 7) * 16u;
    const unsigned ldsb = uni(lds0);
    unsigned m0keep;
#else
    unsigned off0 = stream_pos(0 + (lane >> 3)) + (unsigned)(lane & 7) * 16u, off1 = stream_pos(8 + (lane >> 3)) + (unsigned)(lane & 7) * 16u;
    unsigned off2 = stream_pos(16 + (lane >> 3)) + (unsigned)(lane & 7) * 16u, off3 = stream_pos(24 + (lane >> 3)) + (unsigned)(lane & 7) * 16u;
    const unsigned pc16 = (unsigned)(lane & 7) * 16u, capl = cap * 8u + pc16;
    const unsigned ldsr = lds0 + (unsigned)j * EXACT5_ASM_ROW_BYTES, ldsw = lds0 + (unsigned)(lane >> 3) * EXACT5_ASM_ROW_BYTES + pc16;
#endif
    const long long my_first = (first_seg + j) * (long long)seg_len;
    int rem = (int)max(0ll, min((long long)seg_len, (long long)n_blocks - my_first));
    const long long q_start = q_first + my_first - 32;                                   // < 0: the stream's warm-up reaches before the demodulator's origin
    unsigned ckoff = (rem > 0 && q_start > 0) ? (unsigned)(q_start / kCk) * 8u : 0u;     // an idle or pre-origin stream: any entry of the table (see `hold`)
    // Blocks before the origin (x[i < 0] = 0, phase (1, 0) at block 0: SSBD.hpp:117-121).  Only a channel's first wave can have them, and only while the
    // channel's first pending output is one of the demodulator's first 32: per lane the tiles of its own stream that precede the origin (low half of
    // pk) and, as a loader lane, the load tiles of its first stream that do (high half); the statement zeroes those rows on their way into LDS and holds
    // the stream's phase.  `hold` / `holdlt` are the wave's largest counts (stream 0's): zero for every other wave, which then only branches over this.
    const long long pre0 = 32 - q_first - first_seg * (long long)seg_len;                // stream 0's pre-origin tiles (<= 0: none in this wave)
    int hold = pre0 > 0 ? (int)pre0 : 0, holdlt = hold * D / 16;
    const long long pre_me = 32 - q_first - my_first, pre_ld = 32 - q_first - (first_seg + (lane >> 3)) * (long long)seg_len;
    const unsigned pk = (unsigned)(pre_me > 0 ? pre_me : 0) | ((unsigned)((pre_ld > 0 ? pre_ld : 0) * D / 16) << 16);
    const int st1 = seg_len * D / 2, st2 = 2 * st1, st3 = 3 * st1;                        // load tiles between a loader lane's i-th streams (8 streams apart)
    unsigned outoff = (unsigned)my_first * 4u;
    // the matrix instruction's row i (operand lane i & 31) lands in the half (i >> 2) & 1, register (i & 3) + 4 (i >> 3): give row i tap block
    // n = register + 16 half, so that a lane of the lower half holds tap blocks 0..15 in register order and the upper half 16..31
    unsigned tapoff = (unsigned)((j & 3) + 4 * (j >> 3) + 16 * ((j >> 2) & 1)) * (unsigned)(4 * D);
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
    if (stamp && lane == 0) {
        unsigned long long t_, r_;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(r_)::"memory");
        as_global_rw(clk)[0] = t_;
        as_global_rw(clk)[1] = r_;
    }
#if EXACT5_ASM_DMA
    const float2 *ring1 = ring + 16, *ring2 = ring + 32, *ring3 = ring + 48;            // a load's immediate would move the LDS address too: four ring bases instead
#define EXACT5_STATEMENT(TEXT)                                                                                                                             \
    asm volatile(TEXT                                                                                                                                      \
                 : [off0] "+v"(off0), [off1] "+v"(off1), [off2] "+v"(off2), [off3] "+v"(off3), [rem] "+v"(rem), [outoff] "+v"(outoff), [peak] "=&v"(peak), \
                   [ckoff] "+v"(ckoff), [tapoff] "+v"(tapoff), [esave] "=&s"(esave), [warm] "+s"(warm), [iters] "+s"(iters), [hold] "+s"(hold),            \
                   [holdlt] "+s"(holdlt), [m0keep] "=&s"(m0keep)                                                                                           \
                 : [capl0] "v"(capl0), [capl1] "v"(capl1), [pc160] "v"(pc160), [pc161] "v"(pc161), [ldsr] "v"(ldsr), [ldsw] "v"(ldsw), [fj16] "v"(fj16),   \
                   [pk] "v"(pk), [ldsb] "s"(ldsb),                                                                                                         \
                   [ring0] "s"(ring), [ring1] "s"(ring1), [ring2] "s"(ring2), [ring3] "s"(ring3), [taps] "s"(taps_u), [tone] "s"(tone), [ckpt] "s"(ckpt),  \
                   [out] "s"(out),                                                                                                                         \
                   [incre] "s"(incre), [incim] "s"(incim), [sign] "s"(sign), [nsign] "s"(nsign), [hmask] "s"(hmask), [st1] "s"(st1), [st2] "s"(st2),       \
                   [st3] "s"(st3)                                                                                                                          \
                 : EXACT5_ASM_CLOBBERS)
#else
#define EXACT5_STATEMENT(TEXT)                                                                                                                             \
    asm volatile(TEXT                                                                                                                                      \
                 : [off0] "+v"(off0), [off1] "+v"(off1), [off2] "+v"(off2), [off3] "+v"(off3), [rem] "+v"(rem), [outoff] "+v"(outoff), [peak] "=&v"(peak), \
                   [ckoff] "+v"(ckoff), [tapoff] "+v"(tapoff), [esave] "=&s"(esave), [warm] "+s"(warm), [iters] "+s"(iters), [hold] "+s"(hold),            \
                   [holdlt] "+s"(holdlt)                                                                                                                   \
                 : [capl] "v"(capl), [pc16] "v"(pc16), [ldsr] "v"(ldsr), [ldsw] "v"(ldsw), [pk] "v"(pk),                                                   \
                   [ring] "s"(ring), [taps] "s"(taps_u), [tone] "s"(tone), [ckpt] "s"(ckpt), [out] "s"(out),                                               \
                   [incre] "s"(incre), [incim] "s"(incim), [sign] "s"(sign), [nsign] "s"(nsign), [hmask] "s"(hmask), [st1] "s"(st1), [st2] "s"(st2),       \
                   [st3] "s"(st3)                                                                                                                          \
                 : EXACT5_ASM_CLOBBERS)
#endif
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
#ifndef CWSLG_FIN_NT
#define CWSLG_FIN_NT 0                 // bit 0: non-temporal loads of the float frame, bit 1: non-temporal stores of the int16 frame (A/B: profiles/r5_experiments.txt)
#endif
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
            if (CWSLG_FIN_NT & 1) {                              // the float frame is read once
                a[c] = __builtin_nontemporal_load(reinterpret_cast<const CWSLG_GLOBAL v4f *>(frame + i0));
                b[c] = __builtin_nontemporal_load(reinterpret_cast<const CWSLG_GLOBAL v4f *>(frame + i0 + 4));
            } else {
                a[c] = *reinterpret_cast<const CWSLG_GLOBAL v4f *>(frame + i0);
                b[c] = *reinterpret_cast<const CWSLG_GLOBAL v4f *>(frame + i0 + 4);
            }
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
            if (CWSLG_FIN_NT & 2) __builtin_nontemporal_store(pk, reinterpret_cast<CWSLG_GLOBAL v4u *>(out + i0));
            else *reinterpret_cast<CWSLG_GLOBAL v4u *>(out + i0) = pk;
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

// cwsl_gpu.hip -- host runtime + C ABI of libcwslgpu.so (see include/cwsl_gpu.h).
//
// Reference design  : N Instance threads each pull every IQ block and run a scalar recursive filter
//                     (source/Instance.cpp:178-288).
// This design       : the Receiver pushes each block ONCE into a ring in HBM; one batched kernel advances
//                     ALL channels; one batched kernel finalises ALL frames of a slot-clock group.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC (see cwsl_digi_amd/build.py)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library itself is opened lazily by cwslg_rccl_init (multi_gpu.inc)
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <tuple>
#include <vector>

// -DCWSLG_LAB=1 builds libcwslgpu_lab.so: the measured alternatives of every kernel (round-1/-2 forms, matrix-core and persistent
// variants, memory-traffic probes) and the environment switches that select them.  The product library has ONE kernel per job and
// no environment-dependent arithmetic.
#ifndef CWSLG_LAB
#define CWSLG_LAB 0
#endif

#include "../../include/cwsl_gpu.h"
#include "demod_kernels.hpp"
#if CWSLG_LAB
#include "lab/demod_lab_kernels.hpp"
#endif
#include "host_dsp.hpp"
#include "handoff.hpp"
#include "host/slot_clock.hpp"
#include "host/skimmer_config.hpp"
#include "host/spot_parse.hpp"
#include "sync_kernels.hpp"
#include "ft4sync_kernels.hpp"
#include "longsync_kernels.hpp"

namespace cwslg {

// ---------------------------------------------------------------------------------------------
constexpr int kTile = 256;             // outputs per demod workgroup
constexpr int kTileExact = 512;        // ... of the lab library's demod_exact3_kernel / demod_exact4_kernel (rounds 3-4)
constexpr int kExactThreads = kTileExact / 2;
constexpr int kTileMax = 768;          // the largest tile any demod kernel walks a channel with (sizes the phasor checkpoint tables)
// Fast mode: outputs per workgroup by decimation.  A tile is D (T + 31) samples, so a fixed T = 256 makes the tiles of the lower rates small
// (48 kHz: 9 KB, 360 000 workgroups per 512-slot launch) and the launch dispatch-bound: round 4 measured demod_kernel<4, 256> at 2.03 ms per
// 512 slots -- SLOWER than the exact kernel's 1.44 ms -- and <8, 256> at 2.17 ms.  T grows as D shrinks (the same ~37 KB of IQ per tile at
// 96 kHz; 768 at 48 kHz, where one checkpoint per lane caps the tile at 256 lanes x 4 blocks).
constexpr int fast_tile(int D) { return D == 16 ? 256 : D == 8 ? 512 : 768; }
constexpr int kTileExact2 = 248;       // ... of demod_exact2_kernel, lab build (124 of 128 lanes busy; four tiles of 39.7 KB per CU)
constexpr int kDemodThreads = 256;
constexpr int kFinThreads = 256;
constexpr int kCkptStride = cwslg::kCk;   // blocks between phasor checkpoints
// Pinned staging PER RECEIVER: two halves, allocated at the receiver's first host push and sized by it -- four of that push, between 64 KiB and
// 4 MiB per half (round 3 gave every receiver 2 x 4 MiB: 32 GiB of pinned memory for 4096 private streams; a real-time receiver pushes one
// 16 KiB block at a time and gets 2 x 64 KiB) -- and grown when a larger push arrives.
constexpr size_t kStageHalfMin = 64u << 10, kStageHalfMax = 4u << 20;
constexpr int kWorkBufs = 8;
// Events that only order device work or tell the host that a descriptor buffer is free again: no system-scope fence (a fenced event
// behind the last sync kernel of a slot cost a 0.3 ms L2 write-back before the next launch could start).  Host-visible RESULTS are
// always fetched behind hipStreamSynchronize, which fences.
constexpr unsigned kOrderEvent = hipEventDisableTiming | hipEventDisableSystemFence;
constexpr int kCopyStreams = 4;
constexpr int kFetchStreams = 4;
constexpr unsigned kExact5SegCap = 1408;   // outputs per stream of demod_exact5_kernel at most (launch_exact5)
constexpr unsigned kClkSlots = 1024;      // timed exact-mode demod launches between two drains of the spans

// Host-push staging of one receiver.  The reference has one thread per Receiver (Receiver.hpp:167); each of them gets its own
// pinned double buffer here, filled OUTSIDE the context mutex, so that pushes of different receivers copy in parallel.
struct RxStage {
    std::mutex mu;                     // one push at a time per receiver
    char *h = nullptr;                 // 2 * half, pinned
    size_t half = 0;                   // bytes per half
    size_t pos = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool busy[2] = {false, false};
    ~RxStage()
    {
        if (h) (void)hipHostFree(h);
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }
};

// cwslg_push_iq_many: one pinned buffer per concurrent batch pusher -- [n descriptors][n blocks] -- read by ONE scatter kernel.
constexpr int kBatchStages = 4;
struct BatchStage {
    std::mutex mu;
    char *h = nullptr;                 // pinned, host-mapped
    void *h_dev = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;           // behind the scatter kernel that reads the buffer
    bool busy = false;
};
struct ScatterDesc { float2 *ring; unsigned pos, cap; };     // 16 bytes: where receiver k's block goes

struct Receiver {
    bool open = false;
    std::shared_ptr<RxStage> stage;
    uint32_t fs = 0, iq_len = 0, D = 0;
    int32_t lo_hz = 0;
    float2 *d_ring = nullptr;
    uint32_t cap = 0;                  // complex samples
    uint64_t total = 0;                // samples pushed since open
    std::vector<int> channels;
};

struct PhasorTable {
    float2 *d_ckpt = nullptr;
    size_t n_ckpt = 0;
    int refs = 0;
    bool built = false;
    float2 inc{};
    float2 start = make_float2(1.0f, 0.0f);     // checkpoint 0 (SSBD.hpp:121); the live phase for a table made by Tune(reset = false)
};

struct LongConfig {
    bool enabled = false;
    int nfa_hz = 1400, nfb_hz = 1600;          // jt9 -W ... -L 1400 -H 1600 (DecoderPool.hpp:1033)
    float minsync = 1.2f;                       // fst4_decode's minsync for T/R periods above 15 s
};

struct LongShared {                             // device tables of the 120 s modes' search
    struct Plan { float2 *wa = nullptr, *wn = nullptr, *wb = nullptr, *wfull = nullptr; } p45, p125;
    float2 *d_wspr_T = nullptr, *d_f4w_T = nullptr, *d_w512 = nullptr;
    float *d_win512 = nullptr;
    unsigned char *d_pr3 = nullptr;
    int f4w_jlo = 0, f4w_nband = 0;
};

struct LongChannelBuffers {
    char *d_block = nullptr;
    LongWork w{};
};

struct Channel {
    bool open = false;
    int rx = -1;
    int32_t demod_hz = 0;
    bool usb = true;
    std::string mode;
    int group = 0;
    bool wspr_scale = false;           // mode == "WSPR" exactly (Instance.cpp:320)
    bool sync_ft8 = false, sync_ft4 = false;
    bool sync_wspr = false, sync_fst4w = false;    // 120 s modes with a candidate search (longsync_kernels.hpp)
    LongChannelBuffers longbuf;
    size_t frame_len = 0;
    // device storage (one allocation)
    char *d_block = nullptr;
    float *d_frame[2] = {nullptr, nullptr};
    int16_t *d_i16 = nullptr;
    unsigned *d_peak = nullptr;        // [2]
    float *d_factor = nullptr;
    float2 *d_tone = nullptr;
    // constants
    DemodConstants k;
    // The tuning the channel was opened with.  Instance re-creates its SSBD from its OWN demodFreq / USB after every emitted frame
    // (Instance.cpp:251), so a Tune() on the live object lasts until then.
    DemodConstants k_open;
    int32_t open_demod_hz = 0;
    bool open_usb = true;
    std::tuple<uint32_t, int32_t, int, size_t> phasor_key;
    // Instance state (Instance.cpp:203-276)
    uint64_t fill[2] = {0, 0};
    uint64_t t0[2] = {0, 0};
    int wr = 0, rd = 0;
    int64_t origin_abs = 0;            // receiver sample index at which the demodulator was created
    // SSBD::Tune(F, isUSB, reset = false): the phasor continues from its live value, so the channel walks a PRIVATE checkpoint
    // table (start = that value) until its demodulator is next re-created; and the 32 outputs after the retune point are made by
    // demod_transition_kernel from history mixed with the old tuning (parameters kept by value in `trans`).
    bool use_priv = false;
    PhasorTable priv;
    bool trans_active = false;
    std::shared_ptr<TransWork> trans;  // template of the transition work (tones, phases, sign); positions are filled per launch
    int64_t pend_lo = 0;               // first not-yet-demodulated sample
    uint32_t pend_n = 0;               // samples accepted but not yet demodulated
    uint64_t pend_fill0 = 0;           // frame fill at pend_lo
    bool saturated = false;            // "af buffer full" until the next boundary
    // last finalised frame
    bool have_frame = false;
    int frame_idx = 0;
    uint64_t frame_t0 = 0;
    size_t frame_valid = 0;
    // this boundary's finalise rides in the FT8 spectra kernel (boundary_locked -> sync_launch)
    bool fin_fused = false;
    FinWork fused_fin{};
    size_t i16_end = ~size_t(0);       // d_i16 holds zeros at and beyond this index (the n_valid of the frame it holds); unknown until the first finalise
    // sync results
    uint64_t cand_t0 = 0;              // start epoch of the frame the candidate lists on the device were computed from (0: none yet)
    SyncChannelBuffers syncbuf;
};

struct WorkBuf {
    void *h = nullptr;                 // pinned
    void *h_dev = nullptr;             // the same memory as the device addresses it
    void *d = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
};

struct TimedSpan {
    hipEvent_t a, b;
    int kind;                          // 0 demod, 1 finalize, 2 sync stage (whole), 3 / 4 its FT8 spectra / search + selection kernels
    unsigned gen;                      // cwslg_reset_stats generation it was started in: a span of an older generation is not accounted
};

} // namespace cwslg

using namespace cwslg;

struct cwslg_ctx {
    std::mutex mu;
    int device = 0;
    int cu_count = 256;
    int order_override = 0;            // CWSLG_ITEM_ORDER=1 channel-major, 2 tile-major (A/B); 0 = by topology
    bool exact = true;                 // the default: reference-order arithmetic, frames and candidate lists bit-identical to the reference
                                       // chain's; cwslg_set_exact(ctx, 0) selects the fused polyphase form (faster, within 1e-5 of frame peak)
    bool upload_by_dma = false;        // CWSLG_UPLOAD=dma: descriptors through hipMemcpyAsync as in round 1 (measured alternative)
    int demod_variant = 0;             // CWSLG_DEMOD_VARIANT: 0 = one workgroup per tile (default); measured alternatives: 1 persistent +
                                       // prefetch, 2 persistent loop, 4..7 FIR on the matrix cores (192 kHz); 9..11 memory-traffic probe
    hipStream_t stream = nullptr;
    std::string last_error;            // guarded by err_mu: fail() is also reached from code that does not hold `mu` (the staging copy of a push)
    std::mutex err_mu;
    float scale_ft = 0.90f, scale_wspr = 0.20f;    // CWSL_DIGI.cpp:100-101
    std::vector<Receiver> rxs;
    std::vector<Channel> chans;
    std::map<uint32_t, float *> d_taps;            // per sample rate
    std::map<uint32_t, float *> d_taps2;           // (lab library only) the same taps interleaved for demod_exact3_kernel: [33][D][2] = (h[m + D n], h[m + D (n-1)])
    const char *demod_kernel_name = "";           // the demod kernel the last launch used (cwslg_demod_kernel_name)
    std::map<uint32_t, std::vector<float>> h_taps;
    std::map<std::tuple<uint32_t, int32_t, int, size_t>, PhasorTable> phasors;
    std::vector<decltype(phasors)::key_type> phasor_todo;
    float2 *d_sincos = nullptr;
    // host pushes: H2D copies run on their own stream, ordered against the demod kernels by two events
    hipStream_t copy_stream[kCopyStreams] = {};    // receiver r copies on stream r mod kCopyStreams (several DMA engines side by side)
    hipEvent_t copy_done[kCopyStreams] = {};       // recorded in process_locked: the demod launch waits for every copy enqueued so far
    hipEvent_t demod_done = nullptr;   // recorded on stream after every demod launch: a later copy may overwrite ring history only after it
    bool copies_pending[kCopyStreams] = {};
    bool copy_on_main = true;          // H2D copies on the compute stream: measured 30 GB/s from one pusher thread against 21-28 GB/s on
                                       // the dedicated copy streams (CWSLG_COPY_ON_MAIN=0 selects those: copies then overlap the kernels)
    // frames out: D2H copies of finalised frames run on their own streams behind ONE event recorded after the boundary's kernels (at the
    // first fetch of a frame generation), with the context mutex released -- 4096 fetches do not serialise pushes, launches or each other
    hipStream_t fetch_stream[kFetchStreams] = {};
    hipEvent_t fetch_ev[CWSLG_NUM_GROUPS] = {};          // one per slot-clock group: a group's boundary concerns that group's readers only
    bool fetch_ev_valid[CWSLG_NUM_GROUPS] = {};
    unsigned fetch_rr = 0;
    // Results are handed out whole (Instance.cpp:238-245 copies the frame into the ItemToDecode it pushes, DecoderPool.hpp:174-210): a fetch of
    // a frame or of a candidate list registers here under `mu` before it releases it; the next boundary OF THAT GROUP -- the only writer of the
    // group's d_i16 / d_factor / candidate buffers -- waits, under `mu`, until every registered copy has finished before it queues its kernels.
    // A fetch therefore returns the results of the start_epoch it reports, never the next slot's under it, and a group's fetch_ev is never
    // re-recorded under a waiter.  (Round 6: per group -- an FT8 consumer no longer delays an FT4 boundary -- and candidate lists included.)
    std::mutex fetch_mu;
    std::condition_variable fetch_cv;
    int fetch_inflight[CWSLG_NUM_GROUPS] = {};           // guarded by fetch_mu (incremented with mu held, decremented without)
    std::shared_mutex life_mu;         // shared: a fetch's copy is in flight; exclusive: a close frees device buffers (order: mu, then life_mu)
    BatchStage batch[kBatchStages];
    std::atomic<unsigned> batch_next{0};
    std::atomic<uint64_t> push_calls_a{0}, push_host_ns_a{0};   // cwslg_push_iq's share of stats.push_calls / push_host_ms
    // in-kernel clock of timed exact-mode demod launches: a host-mapped ring of (s_memtime, s_memrealtime) pairs at the start and the end of
    // one wave's / workgroup's life (the demod kernels' `clk` argument), read back by drain_spans
    unsigned long long *clk_h = nullptr, *clk_dev = nullptr;
    unsigned clk_head = 0, clk_tail = 0;
    unsigned exact5_seg_cap = kExact5SegCap, exact5_seg_force = 0;
#ifndef CWSLG_FUSE_FIN_DEFAULT
#define CWSLG_FUSE_FIN_DEFAULT 1       // (-DCWSLG_FUSE_FIN_DEFAULT=0: a measurement build with round 5's separate finalise pass, scripts/gpu_r6_sync_ab.sh)
#endif
#ifndef CWSLG_FUSE_MODE_DEFAULT
#define CWSLG_FUSE_MODE_DEFAULT 1      // 1: the windows are read back from the int16 frame through the cache (product); 2 (lab library only): from an LDS ring
#endif
    int fuse_mode = CWSLG_FUSE_MODE_DEFAULT;
    bool fuse_finalize = CWSLG_FUSE_FIN_DEFAULT != 0;         // FT8 + sync: the slot's finalise inside symbol_spectra_v2_kernel (lab build: CWSLG_FUSE_FIN=0 keeps the separate pass, for A/B)
    int process_min_outputs = 0;       // cwslg_set_process_threshold: 0 every cwslg_process() launches, < 0 the library's own threshold, > 0 that many outputs
    bool use_exact5 = true;            // exact mode: demod_exact5_kernel<D> (lab build: a non-zero CWSLG_DEMOD_VARIANT selects round 3/4's tile kernels instead)
    unsigned stat_gen = 0;             // bumped by cwslg_reset_stats: work timed before a reset is not folded into the figures read after it
    double clk_sum_mhz = 0.0;
    int occ_cache[3][5] = {};          // (lab library) launch_demod: resident demod_exact3 / exact4 workgroups per CU by (D, tile form); per context = per device
    // launch descriptors
    WorkBuf wb[kWorkBufs];
    int wb_next = 0;
    size_t wb_largest = 0;
    // stats / timing
    cwslg_stats stats{};
    bool timing = false;
    std::vector<TimedSpan> spans;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    // sync stage
    SyncConfig sync_cfg;
    SyncShared sync_shared;
    Ft4Tables ft4_tables{};
    bool ft4_dft_valu = false;         // CWSLG_FT4_DFT=valu
    LongConfig long_cfg;
    LongShared long_shared;
    // side stream: optionally (CWSLG_SYNC_VARIANT bit 3) carries the FT8 candidate kernel next to the following demod launch
    hipStream_t side = nullptr;
    hipEvent_t sync2d_done = nullptr, cand_done = nullptr, sync_tail = nullptr;
    bool cand_pending = false;
    int long_variant = 0;              // CWSLG_LONG_VARIANT: bit 0 = FST4W's 125 x 256 stage 1 on the VALU, bit 1 = WSPR's 45 x 1024 stage 1 on the matrix cores
    int sync_variant = 0;              // CWSLG_SYNC_VARIANT: bit mask of measured alternatives in the sync stage (0 = defaults)
    // multi-GPU slot-boundary rendezvous (multi_gpu.inc)
    cwslg_rendezvous_fn rdv_fn = nullptr;
    void *rdv_user = nullptr;
    ncclComm_t rccl_comm = nullptr;
    uint64_t *d_rdv = nullptr, *h_rdv = nullptr;   // [4]: this rank's (frames, group, epoch, flag) = 32 bytes, then [world][4]: every rank's
    int rccl_world = 0;
    std::atomic<uint64_t> rdv_flag{0}, rdv_flags_and{0};   // cwslg_set_rendezvous_flag; AND over the ranks at the last built-in rendezvous
    // cwslg_slot_boundary_begin / _end: the rendezvous of a boundary whose device work is queued but not yet waited for
    bool rdv_pending = false;
    int rdv_group = 0;
    uint64_t rdv_epoch = 0, rdv_mine = 0;
    hipEvent_t rdv_ready = nullptr;                // behind the boundary's finalise (+ sync) kernels on the context stream
};

namespace {

int fail(cwslg_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) {
        std::lock_guard<std::mutex> g(c->err_mu);
        c->last_error = buf;
    }
    return code;
}

#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(ctx, CWSLG_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

int sync_launch(cwslg_ctx *c, const std::vector<int> &emitted);
int sync_ensure_channel(cwslg_ctx *c, Channel &ch);
int long_sync_launch(cwslg_ctx *c, const std::vector<int> &emitted);

WorkBuf *acquire_workbuf(cwslg_ctx *c, size_t bytes)
{
    WorkBuf &w = c->wb[c->wb_next];
    c->wb_next = (c->wb_next + 1) % kWorkBufs;
    if (w.in_flight) {
        hipEventSynchronize(w.done);
        w.in_flight = false;
    }
    if (w.bytes < bytes) {
        // hipFree / hipHostFree wait for the device: a buffer that has to grow costs the host its lead over the GPU (the next launch
        // then starts only after the queue has drained and the host has prepared it: ~0.3 ms of idle GPU per slot at 4096 slots when
        // requests of three sizes rotated over the eight buffers).  So every (re)allocation takes the largest size seen so far and the
        // pool stops growing after its first revolution.
        if (w.h) hipHostFree(w.h);
        if (w.d) hipFree(w.d);
        // (round 6: ... and never less than one descriptor of the largest kind per channel the context has -- SyncWork grew to 112 bytes with the fused
        // finalise and would otherwise arrive, at the first emitting boundary, as a new largest request: eight regrowths, each waiting for the device)
        static_assert(sizeof(SyncWork) <= 128 && sizeof(ChanWork) <= 128 && sizeof(FinWork) <= 128, "per-channel descriptors");
        size_t nb = std::max<size_t>(std::max(std::max(bytes, c->wb_largest), c->chans.size() * (size_t)128), 64 << 10);
        nb = (nb + 4095) & ~size_t(4095);
        c->wb_largest = nb;
        if (hipHostMalloc(&w.h, nb, hipHostMallocDefault) != hipSuccess) return nullptr;
        if (hipHostGetDevicePointer(&w.h_dev, w.h, 0) != hipSuccess) w.h_dev = nullptr;
        if (hipMalloc(&w.d, nb) != hipSuccess) return nullptr;
        w.bytes = nb;
    }
    if (!w.done) hipEventCreateWithFlags(&w.done, kOrderEvent);
    return &w;
}

// Descriptors go host -> device through a COPY KERNEL on the compute queue, not through hipMemcpyAsync: a DMA-engine copy queued
// behind a kernel makes the runtime resolve the cross-engine dependency on the host, which keeps the host from running ahead of the
// GPU -- every launch sequence then starts only after the previous kernel has finished and the GPU idles for the host's preparation
// time (rocprofv3 timeline at 4096 slots: 0.33 ms between the last sync kernel of a slot and the next demod launch).
__global__ void upload_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, unsigned n16)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}
// cwslg_push_iq_many: block k of a batch (n_pairs x 16 bytes, read straight from the host-mapped staging buffer over the host link) goes
// to ring k at its write position, wrapping at the ring's end.  grid (ceil(n_pairs / 256), receivers): every wave-level access is one
// contiguous 1 KiB run on both sides (positions and capacities are multiples of 4 D samples, so a 16-byte pair never straddles the wrap).
__global__ __launch_bounds__(256) void scatter_blocks_kernel(const ScatterDesc *__restrict__ desc, const v4f *__restrict__ src, unsigned n_pairs)
{
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_pairs) return;
    const ScatterDesc d = desc[blockIdx.y];
    const v4f v = src[(size_t)blockIdx.y * n_pairs + i];
    unsigned idx = d.pos + 2u * i;
    if (idx >= d.cap) idx -= d.cap;
    reinterpret_cast<CWSLG_GLOBAL v4f *>((uintptr_t)d.ring)[idx >> 1] = v;     // (a ring is HBM: global_store, not flat_store)
}
hipError_t upload_workbuf(cwslg_ctx *c, WorkBuf *w, size_t bytes)
{
    if (!w->h_dev || c->upload_by_dma) return hipMemcpyAsync(w->d, w->h, bytes, hipMemcpyHostToDevice, c->stream);
    const unsigned n16 = (unsigned)((bytes + 15) / 16);                  // buffers are sized in 4 KB steps
    hipLaunchKernelGGL(upload_kernel, dim3((n16 + 255) / 256), dim3(256), 0, c->stream, (uint4 *)w->d, (const uint4 *)w->h_dev, n16);
    return hipGetLastError();
}

void span_begin(cwslg_ctx *c, int kind, hipEvent_t *a, hipEvent_t *b)
{
    *a = *b = nullptr;
    if (!c->timing) return;
    std::pair<hipEvent_t, hipEvent_t> p;
    if (!c->ev_pool.empty()) {
        p = c->ev_pool.back();
        c->ev_pool.pop_back();
    } else {
        hipEventCreate(&p.first);
        hipEventCreate(&p.second);
    }
    *a = p.first;
    *b = p.second;
    hipEventRecord(*a, c->stream);
    c->spans.push_back({*a, *b, kind, c->stat_gen});
}
void span_end(cwslg_ctx *c, hipEvent_t b)
{
    if (b) hipEventRecord(b, c->stream);
}
// a span on another stream of the context (both events on that stream)
void span_begin_on(cwslg_ctx *c, int kind, hipStream_t st, hipEvent_t *a, hipEvent_t *b)
{
    *a = *b = nullptr;
    if (!c->timing) return;
    std::pair<hipEvent_t, hipEvent_t> p;
    if (!c->ev_pool.empty()) { p = c->ev_pool.back(); c->ev_pool.pop_back(); }
    else { hipEventCreate(&p.first); hipEventCreate(&p.second); }
    *a = p.first; *b = p.second;
    hipEventRecord(*a, st);
    c->spans.push_back({*a, *b, kind, c->stat_gen});
}
// every stream of the context that can hold device work (host-visible results must wait for all of them)
hipError_t sync_streams(cwslg_ctx *c)
{
    if (c->side) { hipError_t e = hipStreamSynchronize(c->side); if (e != hipSuccess) return e; }
    return hipStreamSynchronize(c->stream);
}
// before device buffers are freed: no fetch may still be copying out of them (caller holds the context mutex)
void wait_fetches(cwslg_ctx *c)
{
    std::unique_lock<std::shared_mutex> l(c->life_mu);
    for (hipStream_t fs : c->fetch_stream) if (fs) (void)hipStreamSynchronize(fs);
}
// Only call with the stream idle (after hipStreamSynchronize).
void drain_spans(cwslg_ctx *c)
{
    // spans whose end event has not completed (another thread queued work since the caller's wait) stay for the next drain
    std::vector<TimedSpan> later;
    for (const TimedSpan &s : c->spans) {
        if (hipEventQuery(s.b) != hipSuccess) { later.push_back(s); continue; }
        float ms = 0.f;
        if (s.gen == c->stat_gen && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            if (s.kind == 0) c->stats.demod_ms += ms;
            else if (s.kind == 1) c->stats.finalize_ms += ms;
            else if (s.kind == 3) c->stats.sync_spectra_ms += ms;
            else if (s.kind == 4) c->stats.sync_search_ms += ms;
            else c->stats.sync_ms += ms;
        }
        c->ev_pool.push_back({s.a, s.b});
    }
    c->spans.swap(later);
    // clock slots, oldest first: a slot whose end stamps have landed belongs to a finished launch whatever else is queued (a real-time host
    // never reaches an empty span list); one without them is either still running -- stop there -- or, with nothing queued any more, a
    // launch whose stamping workgroup had no work: skip it
    for (; c->clk_tail != c->clk_head; ++c->clk_tail) {
        const volatile unsigned long long *q = c->clk_h + 4 * (c->clk_tail % kClkSlots);
        if (!(q[2] && q[3]) && !c->spans.empty()) break;
        if (q[0] && q[1] && q[2] > q[0] && q[3] > q[1] + 300) {        // >= 3 us of the 100 MHz counter (a tile workgroup of the fast kernel lives ~6 us)
            c->clk_sum_mhz += 100.0 * (double)(q[2] - q[0]) / (double)(q[3] - q[1]);
            c->stats.demod_clock_launches++;
            c->stats.demod_clock_mhz = c->clk_sum_mhz / (double)c->stats.demod_clock_launches;
        }
    }
}

int ensure_taps(cwslg_ctx *c, uint32_t fs)
{
    if (c->d_taps.count(fs)) return CWSLG_OK;
    std::vector<float> h = design_taps(fs, kSsbBw);
    float *d = nullptr;
    HIPCHK(c, hipMalloc(&d, h.size() * sizeof(float)));
    HIPCHK(c, hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    c->d_taps[fs] = d;
#if CWSLG_LAB
    // (lab library) demod_exact3_kernel / demod_exact4_kernel read the taps as wave-uniform PAIRS: step n of a thread's two adjacent outputs uses tap
    // block n for the even one and n - 1 for the odd one; blocks -1 and 32 do not exist (their addends are masked in the kernel): zeros
    const size_t D = h.size() / 32;
    std::vector<float> h2(33 * D * 2, 0.0f);
    for (size_t n = 0; n < 33; ++n)
        for (size_t m = 0; m < D; ++m) {
            if (n <= 31) h2[(n * D + m) * 2] = h[m + D * n];
            if (n >= 1) h2[(n * D + m) * 2 + 1] = h[m + D * (n - 1)];
        }
    float *d2 = nullptr;
    HIPCHK(c, hipMalloc(&d2, h2.size() * sizeof(float)));
    HIPCHK(c, hipMemcpy(d2, h2.data(), h2.size() * sizeof(float), hipMemcpyHostToDevice));
    c->d_taps2[fs] = d2;
#endif
    c->h_taps[fs] = std::move(h);
    return CWSLG_OK;
}

int launch_phasor_jobs(cwslg_ctx *c, const std::vector<PhasorJob> &jobs)
{
    if (jobs.empty()) return CWSLG_OK;
    WorkBuf *w = acquire_workbuf(c, jobs.size() * sizeof(PhasorJob));
    if (!w) return fail(c, CWSLG_ERR_NOMEM, "work buffer allocation failed");
    std::memcpy(w->h, jobs.data(), jobs.size() * sizeof(PhasorJob));
    HIPCHK(c, upload_workbuf(c, w, jobs.size() * sizeof(PhasorJob)));
    const int n = (int)jobs.size();
    unsigned max_ckpt = 0;
    for (const PhasorJob &j : jobs) max_ckpt = std::max(max_ckpt, j.n_ckpt);
    hipLaunchKernelGGL(phasor_coarse_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, (const PhasorJob *)w->d, n);
    const unsigned segs = (max_ckpt + kCoarse - 1) / kCoarse;
    hipLaunchKernelGGL(phasor_fine_kernel, dim3((segs + 63) / 64, (unsigned)n), dim3(64), 0, c->stream, (const PhasorJob *)w->d);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(w->done, c->stream));
    w->in_flight = true;
    return CWSLG_OK;
}

int build_pending_phasors(cwslg_ctx *c)
{
    if (c->phasor_todo.empty()) return CWSLG_OK;
    std::vector<PhasorJob> jobs;
    for (auto &key : c->phasor_todo) {
        auto it = c->phasors.find(key);
        if (it == c->phasors.end() || it->second.built) continue;
        PhasorJob j{};
        j.ckpt = it->second.d_ckpt;
        j.inc = it->second.inc;
        j.start = it->second.start;
        j.n_ckpt = (unsigned)it->second.n_ckpt;
        jobs.push_back(j);
        it->second.built = true;
    }
    c->phasor_todo.clear();
    return launch_phasor_jobs(c, jobs);
}

// The channel's private checkpoint table (Tune with reset = false): n_ckpt entries from `start` with step `inc`, (re)built on the
// context stream.  Caller holds the mutex; a table that may be in use by a queued launch is only replaced after a stream wait.
int build_private_phasor(cwslg_ctx *c, Channel &ch, size_t n_ckpt, float2 start, float2 inc)
{
    PhasorTable &pt = ch.priv;
    if (pt.d_ckpt && pt.n_ckpt != n_ckpt) {
        HIPCHK(c, sync_streams(c));
        (void)hipFree(pt.d_ckpt);
        pt.d_ckpt = nullptr;
    }
    if (!pt.d_ckpt && hipMalloc(&pt.d_ckpt, n_ckpt * sizeof(float2)) != hipSuccess) {
        pt.d_ckpt = nullptr;
        return fail(c, CWSLG_ERR_NOMEM, "phasor table allocation failed");
    }
    pt.n_ckpt = n_ckpt; pt.start = start; pt.inc = inc; pt.built = true; pt.refs = 1;
    PhasorJob j{};
    j.ckpt = pt.d_ckpt; j.inc = inc; j.start = start; j.n_ckpt = (unsigned)n_ckpt;
    return launch_phasor_jobs(c, std::vector<PhasorJob>{j});
}

void drop_private_phasor(Channel &ch)       // the demodulator is re-created: back to the shared table of its tuning
{
    ch.use_priv = false;
    ch.trans_active = false;
}

int retarget_phasor(cwslg_ctx *c, Channel &ch, const std::tuple<uint32_t, int32_t, int, size_t> &new_key, float2 inc);

// Instance.cpp:251: the new SSBD is constructed from the Instance's own demodFreq and USB, so a channel retuned by Tune() goes
// back to the tuning it was opened with.  Only a retuned channel pays the stream wait (its retuned table may be in use).
int restore_open_tuning(cwslg_ctx *c, Channel &ch, const Receiver &rx)
{
    if (ch.demod_hz == ch.open_demod_hz && ch.usb == ch.open_usb) return CWSLG_OK;
    HIPCHK(c, sync_streams(c));
    const auto key = std::make_tuple(rx.fs, ch.open_demod_hz, ch.open_usb ? 1 : 0, std::get<3>(ch.phasor_key));
    int rc = retarget_phasor(c, ch, key, make_float2(ch.k_open.inc.real(), ch.k_open.inc.imag()));
    if (rc) return rc;
    ch.k = ch.k_open;
    ch.demod_hz = ch.open_demod_hz;
    ch.usb = ch.open_usb;
    HIPCHK(c, hipMemcpyAsync(ch.d_tone, ch.k.tone.data(), ch.k.block * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, sync_streams(c));             // (the source is this channel's own storage; keep it simple on this rare path)
    return CWSLG_OK;
}

// demod_exact5_kernel (192 kHz, exact mode): one wave per (channel, chunk of 32 streams x seg_len outputs).  seg_len trades the 32-block warm-up
// of every stream (32 / seg_len of extra work) against waves to fill the chip with: at least two rounds of resident waves where the work allows.
// Round 6: that choice is the LATENCY policy -- right at a slot boundary, where frames are waited for (with few pending blocks it goes down to 4
// outputs per stream: 36 tiles for 4 outputs, nine times the arithmetic, but every wave slot of the chip busy and the launch as short as it can be).
// A launch that nobody waits for -- cwslg_process() in the middle of a slot, or ring pressure -- takes the THROUGHPUT policy instead: streams as
// long as the channel with the most pending blocks allows with all 32 lanes of its wave busy (max_blocks / 32), i.e. one wave per channel and
// warm-up share 32 / seg, unless the latency policy's streams are longer still (bench scale).  stats.demod_blocks_read counts the blocks a
// launch actually fetches and puts through the arithmetic, warm-up included: demod_blocks_read x D / demod_samples is the redundancy.
// (pure: exported as cwslg_exact_stream_length so that the CPU suite can hold the policy to the figures the notebook quotes)
unsigned exact5_stream_length(uint64_t total_blocks, unsigned max_blocks, unsigned cu_count, unsigned seg_cap, bool latency)
{
    const uint64_t waves_min = (uint64_t)cu_count * 8 * 2;
    unsigned seg = (unsigned)std::min<uint64_t>(seg_cap, (total_blocks + 32 * waves_min - 1) / (32 * waves_min));
    seg = std::max(4u, (seg + 3) / 4 * 4);
    if (!latency) seg = std::max(seg, std::min(seg_cap, max_blocks / 32 / 4 * 4));
    return seg;
}

template <int D>
int launch_exact5(cwslg_ctx *c, const std::vector<ChanWork> &works, unsigned max_blocks, uint32_t fs, bool chunk_major, bool latency)
{
    if (works.empty()) return CWSLG_OK;
    uint64_t total_blocks = 0;
    for (const ChanWork &w : works) total_blocks += w.n_blocks;
    unsigned seg = exact5_stream_length(total_blocks, max_blocks, (unsigned)c->cu_count, c->exact5_seg_cap, latency);
    if (c->exact5_seg_force) seg = c->exact5_seg_force;
    const int chunks = (int)((max_blocks + 32 * seg - 1) / (32 * seg));
    for (const ChanWork &w : works) {                   // streams of seg outputs (the last one shorter), each 32 blocks of warm-up
        const uint64_t streams = ((uint64_t)w.n_blocks + seg - 1) / seg;
        c->stats.demod_blocks_read += w.n_blocks + 32 * streams;
    }
    WorkBuf *w = acquire_workbuf(c, works.size() * sizeof(ChanWork));
    if (!w) return fail(c, CWSLG_ERR_NOMEM, "work buffer allocation failed");
    std::memcpy(w->h, works.data(), works.size() * sizeof(ChanWork));
    HIPCHK(c, upload_workbuf(c, w, works.size() * sizeof(ChanWork)));
    const long long items = (long long)chunks * (long long)works.size();
    hipEvent_t ea, eb;
    span_begin(c, 0, &ea, &eb);
    c->demod_kernel_name = D == 16 ? "demod_exact5_kernel<16>" : D == 8 ? "demod_exact5_kernel<8>" : "demod_exact5_kernel<4>";
    unsigned long long *clk = nullptr;
    if (c->timing && c->clk_dev && c->clk_head - c->clk_tail < kClkSlots) {
        const unsigned slot = c->clk_head++ % kClkSlots;
        std::memset(c->clk_h + 4 * slot, 0, 4 * sizeof(unsigned long long));
        clk = c->clk_dev + 4 * slot;
    }
    hipLaunchKernelGGL(demod_exact5_kernel<D>, dim3((unsigned)((items + kExact5Waves - 1) / kExact5Waves)), dim3(64 * kExact5Waves), 0, c->stream,
                       (const ChanWork *)w->d, (const float *)c->d_taps[fs], chunk_major ? -chunks : chunks, (int)works.size(), (int)seg, clk);
    span_end(c, eb);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(w->done, c->stream));
    w->in_flight = true;
    c->stats.demod_launches++;
    return CWSLG_OK;
}

template <int D>
int launch_demod(cwslg_ctx *c, const std::vector<ChanWork> &works, unsigned max_blocks, uint32_t fs, bool tile_major, bool latency)
{
    if (works.empty()) return CWSLG_OK;
    if (c->exact && c->use_exact5) {
        // exact mode: demod_exact5_kernel<D> takes everything that meets its alignment rules -- which every push through this library does (pushes are
        // whole multiples of 4 blocks, cwslg_push_iq's CWSLG_ERR_BLOCK); anything else is refused below (the lab library still has round 3/4's tile kernels)
        bool aligned = true;
        for (const ChanWork &w : works)
            aligned = aligned && w.q_first >= 0 && w.q_first % 4 == 0 && w.n_blocks % 4 == 0 && w.lo_mod % (4 * D) == 0 && w.ring_cap % (4 * D) == 0 &&
                      (uint64_t)w.ring_cap * 8 < (1ull << 32) - 4096;      // (32-bit byte offsets into the ring, advanced by up to 512 before the wrap test)
        if (aligned) return launch_exact5<D>(c, works, max_blocks, fs, tile_major, latency);
    }
    // the descriptors, then (64-byte aligned) eight per-XCD work counters (the lab library's persistent tile kernels draw from them), zero at launch
    const size_t ctr_off = (works.size() * sizeof(ChanWork) + 63) & ~size_t(63);
    WorkBuf *w = acquire_workbuf(c, ctr_off + 64);
    if (!w) return fail(c, CWSLG_ERR_NOMEM, "work buffer allocation failed");
    std::memcpy(w->h, works.data(), works.size() * sizeof(ChanWork));
    std::memset((char *)w->h + works.size() * sizeof(ChanWork), 0, ctr_off + 64 - works.size() * sizeof(ChanWork));
    HIPCHK(c, upload_workbuf(c, w, ctr_off + 64));
    // product build: ONE kernel per job -- demod_exact5_kernel<D> in exact mode (the default; launched above), demod_kernel in fast mode.  The measured
    // alternatives (CWSLG_DEMOD_VARIANT) exist in the lab build only (-DCWSLG_LAB=1 -> libcwslgpu_lab.so).
    int tile = c->exact ? kTileExact : fast_tile(D);
#if CWSLG_LAB
    if (!c->exact && c->demod_variant != 0) tile = kTile;                       // the measured alternatives of the fast kernel all walk 256-output tiles
    const bool small_tile = !c->exact && c->demod_variant == 15 && D == 16;     // 192-output tiles: 31 KB of LDS, five workgroups per CU
    if (c->exact && c->demod_variant == 20) tile = kTile;                       // round 1's exact kernel: one output per thread
    if (c->exact && c->demod_variant == 21) tile = kTileExact2;
    if (c->exact && c->demod_variant == 23) tile = 256;                         // exact3 with two-wave workgroups (four per CU)
    if (c->exact && c->demod_variant == 24) tile = 128;                         // exact3 with one-wave workgroups (seven per CU)
    if (small_tile) tile = 192;
#endif
    const int tiles_n = (int)((max_blocks + tile - 1) / tile);
    for (const ChanWork &cw : works)                            // a tile of T outputs reads T + 31 blocks (the filter's history)
        c->stats.demod_blocks_read += cw.n_blocks + 31 * (((uint64_t)cw.n_blocks + tile - 1) / tile);
    const int tiles_x = tile_major ? -tiles_n : tiles_n;        // sign selects the work-item order (demod_kernels.hpp)
    const long long total = (long long)tiles_n * (long long)works.size();
    const long long per_xcd = (total + 7) / 8;
    hipEvent_t ea, eb;
    span_begin(c, 0, &ea, &eb);
    bool launched = false;
#if CWSLG_LAB
#include "lab/demod_lab_dispatch.inc"      // CWSLG_DEMOD_VARIANT: the measured alternatives (sets `launched`)
#endif
#if CWSLG_LAB
#include "lab/demod_lab_exact_dispatch.inc" // exact mode through round 3/4's tile kernels (CWSLG_DEMOD_VARIANT 23-27; sets `launched`)
#endif
    if (!launched && c->exact) {
        span_end(c, eb);
        return fail(c, CWSLG_ERR_UNSUPPORTED, "exact mode: a launch that does not meet demod_exact5_kernel's alignment rules (pushes are multiples of 4 blocks)");
    } else if (!launched) {
        c->demod_kernel_name = D == 16 ? "demod_kernel<16,256,256,0>" : D == 8 ? "demod_kernel<8,512,256,0>" : "demod_kernel<4,768,256,0>";
        unsigned long long *clk = nullptr;
        if (c->timing && c->clk_dev && c->clk_head - c->clk_tail < kClkSlots) {
            const unsigned slot = c->clk_head++ % kClkSlots;
            std::memset(c->clk_h + 4 * slot, 0, 4 * sizeof(unsigned long long));
            clk = c->clk_dev + 4 * slot;
        }
        hipLaunchKernelGGL((demod_kernel<D, fast_tile(D), kDemodThreads, 0>), dim3((unsigned)(per_xcd * 8)), dim3(kDemodThreads), 0,
                           c->stream, (const ChanWork *)w->d, (const float *)c->d_taps[fs], tiles_x, (int)works.size(), clk);
    }
    span_end(c, eb);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(w->done, c->stream));
    w->in_flight = true;
    c->stats.demod_launches++;
    return CWSLG_OK;
}


// Point a channel at the phasor table `new_key` (creating and scheduling it if nobody uses it yet) and drop its
// reference to the old one.  Caller holds the mutex and has synchronised the stream if the old table may be in use.
int retarget_phasor(cwslg_ctx *c, Channel &ch, const std::tuple<uint32_t, int32_t, int, size_t> &new_key, float2 inc)
{
    if (new_key == ch.phasor_key) return CWSLG_OK;
    PhasorTable &pt = c->phasors[new_key];
    if (pt.refs == 0) {
        pt.d_ckpt = nullptr;
        if (hipMalloc(&pt.d_ckpt, std::get<3>(new_key) * sizeof(float2)) != hipSuccess) {
            c->phasors.erase(new_key);
            return fail(c, CWSLG_ERR_NOMEM, "phasor table allocation failed");
        }
        pt.n_ckpt = std::get<3>(new_key);
        pt.inc = inc;
        pt.built = false;
        c->phasor_todo.push_back(new_key);
    }
    pt.refs++;
    auto it = c->phasors.find(ch.phasor_key);
    if (it != c->phasors.end() && --it->second.refs == 0) {
        c->phasor_todo.erase(std::remove(c->phasor_todo.begin(), c->phasor_todo.end(), ch.phasor_key), c->phasor_todo.end());
        hipFree(it->second.d_ckpt);
        c->phasors.erase(it);
    }
    ch.phasor_key = new_key;
    return CWSLG_OK;
}

// Checkpoints a launch may touch for blocks [q_first, q_first + n_blocks), whatever tile size T <= kTileMax the kernel walks them with: the
// last tile starts at or before block q_first + n_blocks - 1, is walked whole (T + 31 blocks from its first input block), and every
// lane of the phasor rebuild reads one checkpoint whether it is used or not (NCK = (T + 31 + 3)/4 + 1 of them from the tile's
// first): the highest index is below (q_first + n_blocks + 2 T + 2)/4 + 1.
inline size_t ckpt_need(long long q_first, unsigned n_blocks)
{
    const long long last = std::max<long long>(0, q_first + (long long)n_blocks + 2 * kTileMax + 2);
    return (size_t)(last / kCkptStride) + 4;
}

// Demodulate everything pending.  Caller holds the mutex.  latency: somebody waits for the result (a slot boundary, a retune, a mode switch) --
// see launch_exact5 for what that selects; cwslg_process() and ring pressure pass false.
int process_locked(cwslg_ctx *c, bool latency = true)
{
    int rc = CWSLG_OK;
    // The phasor recurrence restarts only when a frame is EMITTED (Instance.cpp:251).  A channel whose boundaries keep
    // discarding (epoch 0, or a first boundary that arrives late) keeps counting blocks from its creation, so its
    // checkpoint table is grown -- the same serial recurrence walked further, bit-identical -- before any launch
    // could index past it.
    for (size_t r = 0; r < c->rxs.size(); ++r) {
        Receiver &rx = c->rxs[r];
        if (!rx.open) continue;
        for (int id : rx.channels) {
            Channel &ch = c->chans[id];
            if (!ch.open || ch.pend_n == 0) continue;
            const size_t need = ckpt_need((ch.pend_lo - ch.origin_abs) / (int64_t)rx.D, ch.pend_n / rx.D);
            if (ch.use_priv) {
                if (need > ch.priv.n_ckpt) {
                    rc = build_private_phasor(c, ch, std::max(need, 2 * ch.priv.n_ckpt), ch.priv.start, ch.priv.inc);
                    if (rc) return rc;
                    c->stats.phasor_regrows++;
                }
                continue;
            }
            const size_t have = std::get<3>(ch.phasor_key);
            if (need <= have) continue;
            HIPCHK(c, sync_streams(c));             // the old table may be in use by a queued launch
            auto key = ch.phasor_key;
            std::get<3>(key) = std::max(need, 2 * have);
            rc = retarget_phasor(c, ch, key, make_float2(ch.k.inc.real(), ch.k.inc.imag()));
            if (rc) return rc;
            c->stats.phasor_regrows++;
        }
    }
    rc = build_pending_phasors(c);
    if (rc) return rc;
    for (int k = 0; k < kCopyStreams; ++k)
        if (c->copies_pending[k]) {      // every host push enqueued so far lands before the kernels below read the rings
            HIPCHK(c, hipEventRecord(c->copy_done[k], c->copy_stream[k]));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->copy_done[k], 0));
            c->copies_pending[k] = false;
        }
    // one launch per distinct sample rate; channels grouped by receiver so that a receiver's channels are neighbours
    std::map<uint32_t, std::vector<ChanWork>> by_fs;
    std::map<uint32_t, std::vector<TransWork>> trans_by_fs;
    std::map<uint32_t, unsigned> max_blocks;
    std::map<uint32_t, unsigned> max_share;            // most channels with pending work on one receiver
    for (size_t r = 0; r < c->rxs.size(); ++r) {
        Receiver &rx = c->rxs[r];
        if (!rx.open) continue;
        unsigned share = 0;
        for (int id : rx.channels) {
            Channel &ch = c->chans[id];
            if (!ch.open || ch.pend_n == 0) continue;
            if (ch.trans_active) {
                // the 32 outputs after a Tune(reset = false): history mixed with the old tuning (origin_abs is the retune point)
                const int64_t rel = (ch.pend_lo - ch.origin_abs) / (int64_t)rx.D;
                if (rel < 32) {
                    const uint32_t t_n = (uint32_t)std::min<int64_t>(ch.pend_n / rx.D, 32 - rel);
                    TransWork t = *ch.trans;
                    t.ring = rx.d_ring;
                    t.ring_cap = rx.cap;
                    t.pos_b0 = (unsigned)((uint64_t)ch.origin_abs % rx.cap);
                    t.out = ch.d_frame[ch.wr] + ch.pend_fill0;
                    t.peak = ch.d_peak + ch.wr;
                    t.o_first = (int)rel;
                    t.n_out = (int)t_n;
                    trans_by_fs[rx.fs].push_back(t);
                    c->stats.demod_samples += (uint64_t)t_n * rx.D;
                    ch.pend_lo += (int64_t)t_n * rx.D;
                    ch.pend_fill0 += t_n;
                    ch.pend_n -= t_n * rx.D;
                    if (rel + t_n >= 32) { ch.trans_active = false; ch.trans.reset(); }
                } else {
                    ch.trans_active = false; ch.trans.reset();
                }
                if (ch.pend_n == 0) continue;
            }
            ++share;
            ChanWork w{};
            w.ring = rx.d_ring;
            w.out = ch.d_frame[ch.wr] + ch.pend_fill0;
            w.peak = ch.d_peak + ch.wr;
            w.ckpt = ch.use_priv ? ch.priv.d_ckpt : c->phasors[ch.phasor_key].d_ckpt;
            w.tone = ch.d_tone;
            w.ring_cap = rx.cap;
            w.n_blocks = ch.pend_n / rx.D;
            w.inc = make_float2(ch.k.inc.real(), ch.k.inc.imag());
            w.sign = ch.k.sign;
            w.lo_mod = (unsigned)((uint64_t)ch.pend_lo % rx.cap);
            w.q_first = (ch.pend_lo - ch.origin_abs) / (int64_t)rx.D;
            by_fs[rx.fs].push_back(w);
            max_blocks[rx.fs] = std::max(max_blocks[rx.fs], w.n_blocks);
            c->stats.demod_samples += ch.pend_n;
            ch.pend_lo += ch.pend_n;
            ch.pend_fill0 += ch.pend_n / rx.D;
            ch.pend_n = 0;
        }
        max_share[rx.fs] = std::max(max_share[rx.fs], share);
    }
    for (auto &kv : by_fs) {
        const uint32_t fs = kv.first;
        const uint32_t D = fs / kWaveSR;
        // receivers shared by several channels: tile-major order (the IQ tile is fetched once, then served from L2)
        const bool tile_major = c->order_override ? (c->order_override == 2) : (max_share[fs] >= 2);
        if (D == 16) rc = launch_demod<16>(c, kv.second, max_blocks[fs], fs, tile_major, latency);
        else if (D == 8) rc = launch_demod<8>(c, kv.second, max_blocks[fs], fs, tile_major, latency);
        else if (D == 4) rc = launch_demod<4>(c, kv.second, max_blocks[fs], fs, tile_major, latency);
        else rc = fail(c, CWSLG_ERR_UNSUPPORTED, "sample rate %u unsupported", fs);
        if (rc) return rc;
    }
    for (auto &kv : trans_by_fs) {
        const uint32_t fs = kv.first;
        const uint32_t D = fs / kWaveSR;
        WorkBuf *w = acquire_workbuf(c, kv.second.size() * sizeof(TransWork));
        if (!w) return fail(c, CWSLG_ERR_NOMEM, "work buffer allocation failed");
        std::memcpy(w->h, kv.second.data(), kv.second.size() * sizeof(TransWork));
        HIPCHK(c, upload_workbuf(c, w, kv.second.size() * sizeof(TransWork)));
        const dim3 grid((unsigned)kv.second.size());
        const float *taps = (const float *)c->d_taps[fs];
        if (D == 16) hipLaunchKernelGGL(demod_transition_kernel<16>, grid, dim3(64), 0, c->stream, (const TransWork *)w->d, taps);
        else if (D == 8) hipLaunchKernelGGL(demod_transition_kernel<8>, grid, dim3(64), 0, c->stream, (const TransWork *)w->d, taps);
        else if (D == 4) hipLaunchKernelGGL(demod_transition_kernel<4>, grid, dim3(64), 0, c->stream, (const TransWork *)w->d, taps);
        else return fail(c, CWSLG_ERR_UNSUPPORTED, "sample rate %u unsupported", fs);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(w->done, c->stream));
        w->in_flight = true;
    }
    if (!by_fs.empty() || !trans_by_fs.empty()) HIPCHK(c, hipEventRecord(c->demod_done, c->stream));
    return CWSLG_OK;
}

// Account n new samples (already in the ring) to every channel of the receiver, applying the
// reference's per-block frame-overflow guard (Instance.cpp:268-271) with granularity block_len.
void account_push(cwslg_ctx *c, Receiver &rx, uint32_t n, uint32_t block_len)
{
    const uint64_t first = rx.total;
    for (int id : rx.channels) {
        Channel &ch = c->chans[id];
        if (!ch.open) continue;
        uint32_t done = 0;
        // whole blocks in closed form: block j is accepted iff fill + j*(block_len/D) + block_len <= frame_len-1
        if (!ch.saturated && n >= block_len) {
            const uint64_t fill = ch.fill[ch.wr];
            const uint32_t whole = n / block_len;
            const uint32_t per = block_len / rx.D;
            uint32_t take = 0;
            if (fill + block_len <= ch.frame_len - 1)
                take = (uint32_t)std::min<uint64_t>(whole, (ch.frame_len - 1 - block_len - fill) / per + 1);
            if (take) {
                if (ch.pend_n == 0) {
                    ch.pend_lo = (int64_t)first;
                    ch.pend_fill0 = fill;
                }
                ch.pend_n += take * block_len;
                ch.fill[ch.wr] += (uint64_t)take * per;
                done = take * block_len;
            }
            if (take < whole) ch.saturated = true;       // every later block of this slot fails the same test
        }
        while (done < n) {                                // trailing partial block (and the saturated case)
            const uint32_t blk = std::min(block_len, n - done);
            if (!ch.saturated) {
                const uint64_t fill = ch.fill[ch.wr];
                if (fill + blk > ch.frame_len - 1) ch.saturated = true;
            }
            if (ch.saturated) {
                c->stats.blocks_dropped += (n - done + block_len - 1) / block_len;
                break;
            }
            if (ch.pend_n == 0) {
                ch.pend_lo = (int64_t)(first + done);
                ch.pend_fill0 = ch.fill[ch.wr];
            }
            ch.pend_n += blk;
            ch.fill[ch.wr] += blk / rx.D;
            done += blk;
        }
    }
    rx.total += n;
}

// Make room so that writing n samples cannot overwrite history still needed by pending work.
int reserve_ring(cwslg_ctx *c, Receiver &rx, uint32_t n)
{
    const uint32_t hist = 32 * rx.D;
    if ((uint64_t)n + hist > rx.cap) return fail(c, CWSLG_ERR_ARG, "push of %u samples exceeds ring capacity %u", n, rx.cap);
    uint32_t max_pend = 0;
    for (int id : rx.channels) {
        const Channel &ch = c->chans[id];
        if (ch.open) max_pend = std::max(max_pend, ch.pend_n);
    }
    if ((uint64_t)max_pend + n + hist > rx.cap) {
        int rc = process_locked(c, false);      // the reference logs "I/Q buffer is full!" and stalls (Receiver.hpp:222-229)
        if (rc) return rc;
    }
    return CWSLG_OK;
}

// One in-flight copy of a channel's results (see cwslg_ctx::fetch_inflight).
struct FetchTicket {
    cwslg_ctx *c = nullptr;
    int group = 0;
    void take(cwslg_ctx *ctx, int g_) { std::lock_guard<std::mutex> g(ctx->fetch_mu); ++ctx->fetch_inflight[g_]; c = ctx; group = g_; }
    ~FetchTicket()
    {
        if (!c) return;
        { std::lock_guard<std::mutex> g(c->fetch_mu); --c->fetch_inflight[group]; }
        c->fetch_cv.notify_all();
    }
};

// What a reader of one channel's results holds while it copies them out WITHOUT the context mutex: a ticket of the channel's group (the group's
// next boundary waits for it), the buffers' lifetime, and a fetch stream ordered behind the group's generation event.  Declaration order
// matters: members are destroyed in reverse, so the ticket is released BEFORE `life` -- safe, because the context itself outlives both
// (cwslg_destroy takes life_mu exclusively before it frees anything), and notify_all touches only the context.
struct ResultFetch {
    std::shared_lock<std::shared_mutex> life;
    FetchTicket ticket;
    hipStream_t fs = nullptr;
    hipEvent_t ev = nullptr;
};
// Caller holds c->mu.  ONE event per group and generation of results, recorded behind everything queued so far (the boundary's finalise and
// sync kernels); every fetch of the generation waits for it on a fetch stream, not for the compute stream.
int begin_result_fetch(cwslg_ctx *c, const Channel &ch, ResultFetch &rf)
{
    const int g = ch.group;
    if (!c->fetch_ev_valid[g]) {
        if (c->cand_pending) {                  // (lab variants only) candidate kernels queued on the side stream: the event follows them
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->cand_done, 0));
            c->cand_pending = false;
        }
        HIPCHK(c, hipEventRecord(c->fetch_ev[g], c->stream));
        c->fetch_ev_valid[g] = true;
    }
    rf.fs = c->fetch_stream[c->fetch_rr++ % kFetchStreams];
    rf.ev = c->fetch_ev[g];
    rf.ticket.take(c, g);
    rf.life = std::shared_lock<std::shared_mutex>(c->life_mu);      // the buffers stay allocated until the copy is done
    return CWSLG_OK;
}

int boundary_locked(cwslg_ctx *c, const std::vector<int> &ids, uint64_t epoch_s, uint64_t *n_emitted = nullptr)
{
    if (n_emitted) *n_emitted = 0;
    if (ids.empty()) return CWSLG_OK;
    int rc = process_locked(c);          // everything pushed so far belongs to the finishing slot
    if (rc) return rc;
    std::vector<FinWork> fin;
    std::vector<int> emitted;
    size_t max_len = 0;
    for (int id : ids) {
        Channel &ch = c->chans[id];
        Receiver &rx = c->rxs[ch.rx];
        const int nxt = ch.wr ^ 1;                      // get_next_write_index, ring of 2
        ch.fill[nxt] = 0;                               // memset + reset (Instance.cpp:213-214)
        ch.t0[nxt] = epoch_s;                           // :215
        ch.wr = nxt;                                    // :217
        const int cur = ch.rd;                          // :221 pop_ref
        ch.rd ^= 1;
        ch.saturated = false;
        FinWork f{};
        f.frame = ch.d_frame[cur];
        f.out = ch.d_i16;
        f.peak = ch.d_peak + cur;
        f.peak_next = ch.d_peak + nxt;
        f.factor_out = ch.d_factor;
        f.scale = ch.wspr_scale ? c->scale_wspr : c->scale_ft;
        f.n_valid = (unsigned)ch.fill[cur];
        f.frame_len = (unsigned)ch.frame_len;
        f.emit = (ch.t0[cur] != 0) ? 1 : 0;             // :224-227
        f.tail_end = (unsigned)std::min<size_t>(ch.frame_len, std::max<size_t>(ch.i16_end, ch.fill[cur]));
        if (f.emit) ch.i16_end = ch.fill[cur];          // (either form of the finalise leaves zeros from n_valid on)
        if (f.emit) {
            ch.have_frame = true;
            ch.frame_idx = cur;
            ch.frame_t0 = ch.t0[cur];
            ch.frame_valid = ch.fill[cur];
            ch.origin_abs = (int64_t)rx.total;          // :251 new SSBD: history and phasor restart, the Instance's own tuning
            drop_private_phasor(ch);
            if ((rc = restore_open_tuning(c, ch, rx)) != CWSLG_OK) return rc;
            c->stats.frames_emitted++;
            emitted.push_back(id);
        } else {
            c->stats.frames_discarded++;                // NOTE: no demodulator restart on this path (:226 `continue`)
        }
        ch.pend_fill0 = 0;
        // FT8 channels whose frame goes through the sync stage: symbol_spectra_v2_kernel converts the frame itself (sync_kernels.hpp,
        // spectra_finalize_span) -- no separate memory pass; everything else (no sync, FT4, the 120 s modes, discarded frames) is finalised here
        ch.fin_fused = c->fuse_finalize && f.emit && c->sync_cfg.enabled && ch.sync_ft8 && c->sync_variant == 0 &&
                       ch.frame_len >= (size_t)(FT8_NSTEP * (FT8_NHSYM - 1) + FT8_NSPS);
        if (ch.fin_fused) { ch.fused_fin = f; continue; }
        fin.push_back(f);
        max_len = std::max(max_len, ch.frame_len);
    }
    if (c->cand_pending && (c->sync_variant & 32)) {     // the previous boundary's sync chain (side stream) reads the int16 frames
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->cand_done, 0));
        c->cand_pending = false;
    }
    bool in_group[CWSLG_NUM_GROUPS] = {};
    for (int id : ids) in_group[c->chans[id].group] = true;
    {   // the kernels below rewrite d_i16 / d_factor / the candidate buffers of these groups: let the copies of the previous generation finish
        // first (each is one frame's or one list's D2H copy; new fetches wait for `mu`, which this thread holds)
        std::unique_lock<std::mutex> lk(c->fetch_mu);
        c->fetch_cv.wait(lk, [&] {
            for (int g = 0; g < CWSLG_NUM_GROUPS; ++g) if (in_group[g] && c->fetch_inflight[g] != 0) return false;
            return true;
        });
    }
    if (!fin.empty()) {
        WorkBuf *w = acquire_workbuf(c, fin.size() * sizeof(FinWork));
        if (!w) return fail(c, CWSLG_ERR_NOMEM, "work buffer allocation failed");
        std::memcpy(w->h, fin.data(), fin.size() * sizeof(FinWork));
        HIPCHK(c, upload_workbuf(c, w, fin.size() * sizeof(FinWork)));
        const unsigned gx = (unsigned)((max_len + kFinThreads * 8 * kFinChunks - 1) / (kFinThreads * 8 * kFinChunks));
        hipEvent_t ea, eb;
        span_begin(c, 1, &ea, &eb);
        hipLaunchKernelGGL((finalize_kernel<kFinThreads>), dim3(gx, (unsigned)fin.size()), dim3(kFinThreads), 0, c->stream,
                           (const FinWork *)w->d);
        span_end(c, eb);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(w->done, c->stream));
        w->in_flight = true;
        c->stats.finalize_launches++;
    }
    if (n_emitted) *n_emitted = emitted.size();
    for (int g = 0; g < CWSLG_NUM_GROUPS; ++g)      // a new generation of results: the next fetch records the group's event behind this boundary's kernels
        if (in_group[g]) c->fetch_ev_valid[g] = false;
    // optional sync stage on the freshly finalised int16 frames
    if (c->sync_cfg.enabled && !emitted.empty()) {
        rc = sync_launch(c, emitted);
        if (rc) return rc;
    }
    if (c->long_cfg.enabled && !emitted.empty()) {
        rc = long_sync_launch(c, emitted);
        if (rc) return rc;
    }
    return CWSLG_OK;
}

// A failed rendezvous: the built-in form has already recorded WHAT failed (which rank was at which group / epoch) -- keep that text and
// put the generic line in front of it; a callback's failure has no text of its own.
int rendezvous_failed(cwslg_ctx *c, int rc)
{
    std::string inner;
    {
        std::lock_guard<std::mutex> g(c->err_mu);
        if (c->last_error.rfind("slot-boundary rendezvous:", 0) == 0) inner = c->last_error;
    }
    if (!inner.empty()) return fail(c, rc, "slot-boundary rendezvous failed (%d): %s", rc, inner.c_str() + sizeof("slot-boundary rendezvous:"));
    return fail(c, rc, "slot-boundary rendezvous failed (%d)", rc);
}

} // namespace

// sync stage glue (kept in its own file so this one stays readable)
#include "sync_host.inc"
// multi-GPU rendezvous: callback hook + the built-in RCCL form
#include "multi_gpu.inc"
// candidate search of the 120 s modes (WSPR, FST4W-120)
#include "longsync_host.inc"

// =============================================================================================
extern "C" {

int cwslg_abi_version(void) { return CWSLG_ABI_VERSION; }

const char *cwslg_strerror(int s)
{
    switch (s) {
    case CWSLG_OK: return "ok";
    case CWSLG_ERR_RATIO: return "Fs/B must be an even integer >= 4";
    case CWSLG_ERR_BAND_LOW: return "Signal outside of band (low)";
    case CWSLG_ERR_BAND_HIGH: return "Signal outside of band (high)";
    case CWSLG_ERR_NOMEM: return "out of memory";
    case CWSLG_ERR_MODE: return "Unhandled mode";
    case CWSLG_ERR_ARG: return "invalid argument";
    case CWSLG_ERR_NO_DEVICE: return "no usable HIP device (gfx950 required; there is no CPU fallback)";
    case CWSLG_ERR_HIP: return "HIP runtime error";
    case CWSLG_ERR_NO_FRAME: return "no finalised frame";
    case CWSLG_ERR_UNSUPPORTED: return "unsupported sample rate";
    case CWSLG_ERR_BLOCK: return "block length must be a multiple of SSBD::GetInSize()";
    default: return "unknown status";
    }
}

const char *cwslg_last_error(cwslg_ctx *ctx)
{
    if (!ctx) return "";
    // a per-thread copy: the returned pointer stays valid until this thread asks again, whatever other threads report meanwhile
    static thread_local std::string copy;
    std::lock_guard<std::mutex> g(ctx->err_mu);
    copy = ctx->last_error;
    return copy.c_str();
}

int cwslg_create(cwslg_ctx **out, int device_ordinal)
{
    if (!out) return CWSLG_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return CWSLG_ERR_NO_DEVICE;
    if (device_ordinal < 0) {
        const char *lr = std::getenv("LOCAL_RANK");
        device_ordinal = lr ? std::atoi(lr) % n : 0;
    }
    if (device_ordinal >= n) return CWSLG_ERR_NO_DEVICE;
    if (hipSetDevice(device_ordinal) != hipSuccess) return CWSLG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return CWSLG_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return CWSLG_ERR_NO_DEVICE;   // code object is gfx950-only
    std::unique_ptr<cwslg_ctx, void (*)(cwslg_ctx *)> c(new cwslg_ctx, cwslg_destroy);   // a failed create releases what it made
    c->device = device_ordinal;
    c->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
#if CWSLG_LAB
    // lab build only: the switches that select a measured alternative.  The product library reads no environment variable that
    // changes a kernel or the order of its arithmetic.
    if (const char *v = std::getenv("CWSLG_DEMOD_VARIANT")) c->demod_variant = std::atoi(v);
    if (c->demod_variant != 0) c->use_exact5 = false;          // every measured alternative is an alternative to the tile-shaped launch (26 / 27: round 3's / round 4's exact kernels themselves, for A/B)
    if (const char *v = std::getenv("CWSLG_EXACT5_SEG")) c->exact5_seg_cap = (unsigned)std::max(4, std::atoi(v));        // at most this many outputs per stream
    if (const char *v = std::getenv("CWSLG_EXACT5_SEG_FORCE")) c->exact5_seg_force = (unsigned)std::max(4, std::atoi(v)) / 4 * 4;   // exactly this many (tests)
    if (const char *v = std::getenv("CWSLG_UPLOAD")) c->upload_by_dma = std::strcmp(v, "dma") == 0;
    if (const char *v = std::getenv("CWSLG_FT4_DFT")) c->ft4_dft_valu = std::strcmp(v, "valu") == 0;
    if (const char *v = std::getenv("CWSLG_ITEM_ORDER")) c->order_override = std::atoi(v);
    if (const char *v = std::getenv("CWSLG_SYNC_VARIANT")) c->sync_variant = std::atoi(v);
    if (const char *v = std::getenv("CWSLG_LONG_VARIANT")) c->long_variant = std::atoi(v);
    if (const char *v = std::getenv("CWSLG_COPY_ON_MAIN")) c->copy_on_main = std::atoi(v) != 0;
    if (const char *v = std::getenv("CWSLG_FUSE_FIN")) c->fuse_finalize = std::atoi(v) != 0;
    if (const char *v = std::getenv("CWSLG_FUSE_MODE")) c->fuse_mode = std::atoi(v) == 2 ? 2 : 1;
#endif
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return CWSLG_ERR_HIP;
    for (int k = 0; k < kCopyStreams; ++k) {
        if (hipStreamCreateWithFlags(&c->copy_stream[k], hipStreamNonBlocking) != hipSuccess) return CWSLG_ERR_HIP;
        hipEventCreateWithFlags(&c->copy_done[k], hipEventDisableTiming);
    }
    hipEventCreateWithFlags(&c->demod_done, kOrderEvent);
    for (int k = 0; k < kFetchStreams; ++k)
        if (hipStreamCreateWithFlags(&c->fetch_stream[k], hipStreamNonBlocking) != hipSuccess) return CWSLG_ERR_HIP;
    for (hipEvent_t &e : c->fetch_ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);      // WITH a system-scope fence: the copy engines read what the kernels wrote
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) return CWSLG_ERR_HIP;
    hipEventCreateWithFlags(&c->sync2d_done, hipEventDisableTiming);
    hipEventCreateWithFlags(&c->cand_done, hipEventDisableTiming);
    hipEventCreateWithFlags(&c->sync_tail, hipEventDisableTiming);
    // tone table of the synthetic source (same construction as the oracle's)
    {
        std::vector<float2> tab(4096);
        constexpr double pi = 3.14159265358979323846;
        for (int j = 0; j < 4096; ++j)
            tab[j] = make_float2((float)std::cos(2.0 * pi * j / 4096.0), (float)std::sin(2.0 * pi * j / 4096.0));
        if (hipMalloc(&c->d_sincos, tab.size() * sizeof(float2)) != hipSuccess) return CWSLG_ERR_NOMEM;
        if (hipMemcpy(c->d_sincos, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return CWSLG_ERR_HIP;
    }
    *out = c.release();
    return CWSLG_OK;
}

void cwslg_destroy(cwslg_ctx *c)
{
    if (!c) return;
    hipSetDevice(c->device);
    for (hipStream_t cs : c->copy_stream) if (cs) hipStreamSynchronize(cs);
    if (c->side) hipStreamSynchronize(c->side);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (hipStream_t fs : c->fetch_stream) if (fs) (void)hipStreamSynchronize(fs);
    for (Channel &ch : c->chans) {
        if (ch.d_block) hipFree(ch.d_block);
        sync_free_channel(ch.syncbuf);
        if (ch.longbuf.d_block) { (void)hipFree(ch.longbuf.d_block); ch.longbuf = LongChannelBuffers(); }
        if (ch.priv.d_ckpt) { (void)hipFree(ch.priv.d_ckpt); ch.priv.d_ckpt = nullptr; }
    }
    for (Receiver &rx : c->rxs) if (rx.d_ring) hipFree(rx.d_ring);
    for (auto &kv : c->d_taps) hipFree(kv.second);
    for (auto &kv : c->d_taps2) hipFree(kv.second);
    for (auto &kv : c->phasors) if (kv.second.d_ckpt) hipFree(kv.second.d_ckpt);
    for (WorkBuf &w : c->wb) {
        if (w.h) hipHostFree(w.h);
        if (w.d) hipFree(w.d);
        if (w.done) hipEventDestroy(w.done);
    }
    drain_spans(c);
    for (auto &p : c->ev_pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    rccl_release(c);
    sync_free_shared(c->sync_shared);
    long_free_shared(c->long_shared);
    if (c->d_sincos) hipFree(c->d_sincos);
    if (c->clk_h) (void)hipHostFree(c->clk_h);
    for (hipStream_t fs : c->fetch_stream) if (fs) { (void)hipStreamSynchronize(fs); (void)hipStreamDestroy(fs); }
    for (hipEvent_t e : c->fetch_ev) if (e) (void)hipEventDestroy(e);
    for (BatchStage &b : c->batch) {
        if (b.h) (void)hipHostFree(b.h);
        if (b.ev) (void)hipEventDestroy(b.ev);
    }
    for (int k = 0; k < kCopyStreams; ++k) {
        if (c->copy_stream[k]) { hipStreamSynchronize(c->copy_stream[k]); hipStreamDestroy(c->copy_stream[k]); }
        if (c->copy_done[k]) hipEventDestroy(c->copy_done[k]);
    }
    if (c->demod_done) hipEventDestroy(c->demod_done);
    if (c->rdv_ready) hipEventDestroy(c->rdv_ready);
    if (c->side) { hipStreamSynchronize(c->side); hipStreamDestroy(c->side); }
    for (hipEvent_t e : {c->sync2d_done, c->cand_done, c->sync_tail}) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int cwslg_set_scale_factors(cwslg_ctx *c, float ft, float wspr)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    c->scale_ft = ft;
    c->scale_wspr = wspr;
    return CWSLG_OK;
}

int cwslg_set_exact(cwslg_ctx *c, int on)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    hipSetDevice(c->device);
    int rc = process_locked(c);         // pending samples keep the mode they were pushed under
    if (rc) return rc;
    c->exact = on != 0;
    return CWSLG_OK;
}

int cwslg_receiver_open(cwslg_ctx *c, uint32_t fs, uint32_t iq_len, int32_t lo_hz, uint32_t ring_blocks, int *rx_id)
{
    if (!c || !rx_id || iq_len == 0) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    hipSetDevice(c->device);
    int rc = check_tuning(fs, kSsbBw, 0.0, true);
    if (rc == CWSLG_ERR_RATIO) return fail(c, rc, "Fs/B must be an even integer >= 4");
    const uint32_t D = fs / kWaveSR;
    if (!(D == 4 || D == 8 || D == 16) || D * kWaveSR != fs) return fail(c, CWSLG_ERR_UNSUPPORTED, "sample rate %u: only 48/96/192 kHz", fs);
    if (iq_len % (4 * D) != 0) return fail(c, CWSLG_ERR_BLOCK, "iq_len %u is not a multiple of %u", iq_len, 4 * D);
    rc = ensure_taps(c, fs);
    if (rc) return rc;
    Receiver rx;
    rx.fs = fs; rx.iq_len = iq_len; rx.lo_hz = lo_hz; rx.D = D;
    const uint64_t blocks = ring_blocks ? ring_blocks : (uint64_t)(fs / iq_len + 1) * 3;   // Receiver.hpp:132
    uint64_t cap = blocks * iq_len;
    const uint64_t min_cap = 2ull * (D * (kTileMax + 31)) + 64;    // the tile loader wraps at most once (whatever tile the mode's kernel walks: kTileMax)
    if (cap < min_cap) cap = (min_cap + iq_len - 1) / iq_len * iq_len;
    // demod_exact5_kernel addresses the ring with 32-bit byte offsets: 512 M samples (44 minutes at 192 kHz) less a margin
    if (cap * sizeof(float2) >= (1ull << 32) - 4096) return fail(c, CWSLG_ERR_ARG, "ring too large (%llu samples: a ring is shorter than 4 GiB)", (unsigned long long)cap);
    rx.cap = (uint32_t)cap;
    HIPCHK(c, hipMalloc(&rx.d_ring, (size_t)rx.cap * sizeof(float2)));
    rx.open = true;
    int id = -1;
    for (size_t k = 0; k < c->rxs.size(); ++k) if (!c->rxs[k].open) { id = (int)k; break; }
    if (id < 0) { c->rxs.push_back(rx); id = (int)c->rxs.size() - 1; } else c->rxs[id] = rx;
    *rx_id = id;
    return CWSLG_OK;
}

int cwslg_receiver_close(cwslg_ctx *c, int rx_id)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (rx_id < 0 || rx_id >= (int)c->rxs.size() || !c->rxs[rx_id].open) return fail(c, CWSLG_ERR_ARG, "bad receiver id");
    hipSetDevice(c->device);
    for (hipStream_t cs : c->copy_stream) HIPCHK(c, hipStreamSynchronize(cs));
    HIPCHK(c, sync_streams(c));
    wait_fetches(c);
    Receiver &rx = c->rxs[rx_id];
    for (int id : rx.channels) {        // Receiver::finish terminates its instances (Receiver.hpp:194-199)
        Channel &ch = c->chans[id];
        if (!ch.open) continue;
        if (ch.d_block) hipFree(ch.d_block);
        sync_free_channel(ch.syncbuf);
        if (ch.longbuf.d_block) { (void)hipFree(ch.longbuf.d_block); ch.longbuf = LongChannelBuffers(); }
        auto it = c->phasors.find(ch.phasor_key);
        if (it != c->phasors.end() && --it->second.refs == 0) { hipFree(it->second.d_ckpt); c->phasors.erase(it); }
        if (ch.priv.d_ckpt) (void)hipFree(ch.priv.d_ckpt);
        ch = Channel();
    }
    hipFree(rx.d_ring);
    rx = Receiver();
    return CWSLG_OK;
}

static int push_prologue(cwslg_ctx *c, int rx_id, uint32_t n, Receiver **out)
{
    if (rx_id < 0 || rx_id >= (int)c->rxs.size() || !c->rxs[rx_id].open) return fail(c, CWSLG_ERR_ARG, "bad receiver id");
    Receiver &rx = c->rxs[rx_id];
    if (n == 0 || n % (4 * rx.D) != 0) return fail(c, CWSLG_ERR_BLOCK, "n_complex %u is not a multiple of %u", n, 4 * rx.D);
    hipSetDevice(c->device);
    int rc = reserve_ring(c, rx, n);
    if (rc) return rc;
    *out = &rx;
    return CWSLG_OK;
}

static int push_iq_piece(cwslg_ctx *c, int rx_id, const float *iq, uint32_t n);

int cwslg_push_iq(cwslg_ctx *c, int rx_id, const float *iq, uint32_t n)
{
    if (!c || !iq) return CWSLG_ERR_ARG;
    uint32_t piece = 0;
    std::shared_ptr<RxStage> stage;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (rx_id < 0 || rx_id >= (int)c->rxs.size() || !c->rxs[rx_id].open) return fail(c, CWSLG_ERR_ARG, "bad receiver id");
        Receiver &r = c->rxs[rx_id];
        // a push larger than the ring is fed through it in ring-sized pieces (whole Receiver blocks)
        piece = std::max<uint32_t>(r.iq_len, (r.cap / 2) / r.iq_len * r.iq_len);
        if (!r.stage) r.stage = std::make_shared<RxStage>();
        stage = r.stage;
    }
    std::lock_guard<std::mutex> gp(stage->mu);            // the reference has ONE thread per Receiver; a second pusher waits here
    const auto t_in = std::chrono::steady_clock::now();
    uint32_t done = 0;
    while (done < n) {
        const uint32_t m = std::min(piece, n - done);
        int rc = push_iq_piece(c, rx_id, iq + 2 * (size_t)done, m);
        if (rc) return rc;
        done += m;
    }
    // (atomics, folded into the stats when they are read: no third acquisition of the context mutex per push)
    c->push_calls_a.fetch_add(1, std::memory_order_relaxed);
    c->push_host_ns_a.fetch_add((uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_in).count(), std::memory_order_relaxed);
    return CWSLG_OK;
}

// One piece (<= half the ring).  Three steps: (1) under the context mutex, make room in the ring and read the write position;
// (2) with NO context lock, copy the caller's block (valid only during this call, like a ring slot of the reference) into this
// receiver's pinned staging; (3) under the mutex again, enqueue the H2D copies on the copy stream and account the samples to the
// receiver's channels.  Step 2 is the expensive one (a host memcpy at ~10 GB/s per thread) and runs in parallel across receivers.
static int push_iq_piece(cwslg_ctx *c, int rx_id, const float *iq, uint32_t n)
{
    std::shared_ptr<RxStage> stage;
    uint64_t total0 = 0;
    uint32_t cap = 0;
    float2 *d_ring = nullptr;
    {
        std::lock_guard<std::mutex> g(c->mu);
        Receiver *rx = nullptr;
        int rc = push_prologue(c, rx_id, n, &rx);
        if (rc) return rc;
        stage = rx->stage;
        total0 = rx->total; cap = rx->cap; d_ring = rx->d_ring;
    }
    RxStage &st = *stage;
    hipSetDevice(c->device);
    {
        size_t want = kStageHalfMin;
        while (want < kStageHalfMax && want < 4 * (size_t)n * sizeof(float2)) want <<= 1;
        if (want > st.half) {              // first push, or a larger one than any before: (re)allocate once the copies in flight have drained
            for (int k = 0; k < 2; ++k)
                if (st.busy[k]) { HIPCHK(c, hipEventSynchronize(st.ev[k])); st.busy[k] = false; }
            if (st.h) {                    // copies of segments that never left a half carry no event: drain the stream they were queued on
                HIPCHK(c, hipStreamSynchronize(c->copy_on_main ? c->stream : c->copy_stream[rx_id % kCopyStreams]));
                (void)hipHostFree(st.h);
                st.h = nullptr;
            }
            if (hipHostMalloc((void **)&st.h, 2 * want, hipHostMallocDefault) != hipSuccess) { st.half = 0; return fail(c, CWSLG_ERR_NOMEM, "staging allocation failed"); }
            st.half = want;
            st.pos = 0;
            if (!st.ev[0]) hipEventCreateWithFlags(&st.ev[0], hipEventDisableTiming);
            if (!st.ev[1]) hipEventCreateWithFlags(&st.ev[1], hipEventDisableTiming);
        }
    }
    const size_t kStageHalf = st.half;
    struct Seg { size_t stage_off; uint32_t ring_pos, count; int leaves_half; };
    std::vector<Seg> segs;
    uint32_t done = 0;
    const size_t pos0 = st.pos;                 // where this piece's staging starts (restored if the piece is refused below)
    while (done < n) {
        const size_t half_samples = kStageHalf / sizeof(float2);
        const int half = (int)(st.pos / kStageHalf);
        const size_t off_in_half = st.pos % kStageHalf;
        const size_t room = (kStageHalf - off_in_half) / sizeof(float2);
        if (off_in_half == 0 && st.busy[half]) {
            if (!segs.empty()) break;                       // flush what is staged before waiting for this half to drain
            HIPCHK(c, hipEventSynchronize(st.ev[half]));
            st.busy[half] = false;
        }
        uint32_t m = (uint32_t)std::min<size_t>({(size_t)(n - done), room, half_samples});
        const uint32_t ring_pos = (uint32_t)((total0 + done) % cap);
        m = std::min(m, cap - ring_pos);                    // split at the ring wrap
        std::memcpy(st.h + st.pos, iq + 2 * (size_t)done, (size_t)m * sizeof(float2));
        Seg sg{st.pos, ring_pos, m, -1};
        st.pos += (size_t)m * sizeof(float2);
        if (st.pos % kStageHalf == 0) {                     // leaving a half: fence it once its copy is enqueued
            sg.leaves_half = half;
            if (st.pos == 2 * kStageHalf) st.pos = 0;
        }
        segs.push_back(sg);
        done += m;
    }
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (rx_id >= (int)c->rxs.size() || !c->rxs[rx_id].open || c->rxs[rx_id].d_ring != d_ring)
            return fail(c, CWSLG_ERR_ARG, "receiver closed during a push");
        Receiver &rx = c->rxs[rx_id];
        // the write position was read in step 1: another push on this receiver (a cwslg_push_iq_many batch, or a second thread -- the contract
        // is one pusher per receiver, Receiver.hpp:167) has moved it since, so the staged block would land on top of that push's samples
        if (rx.total != total0) {
            // nothing of THIS piece was enqueued: give its staging bytes back, so that no half counts as "left" without its fence event having
            // been recorded (the halves' reuse discipline stays intact); earlier pieces of the same call are already in the ring and accounted
            st.pos = pos0;
            return fail(c, CWSLG_ERR_ARG, "receiver %d was pushed from two threads at once; this piece of the call was not accounted (earlier pieces of the same call may be)", rx_id);
        }
        // ring history that a queued demod launch still reads must not be overwritten under it
        const int cs = rx_id % kCopyStreams;
        hipStream_t cstr = c->copy_on_main ? c->stream : c->copy_stream[cs];
        if (!c->copy_on_main) HIPCHK(c, hipStreamWaitEvent(cstr, c->demod_done, 0));
        for (const Seg &sg : segs) {
            HIPCHK(c, hipMemcpyAsync(d_ring + sg.ring_pos, st.h + sg.stage_off, (size_t)sg.count * sizeof(float2), hipMemcpyHostToDevice, cstr));
            if (sg.leaves_half >= 0) {
                HIPCHK(c, hipEventRecord(st.ev[sg.leaves_half], cstr));
                st.busy[sg.leaves_half] = true;
            }
        }
        if (!c->copy_on_main) c->copies_pending[cs] = true;
        c->stats.h2d_bytes += (uint64_t)done * sizeof(float2);
        account_push(c, rx, done, rx.iq_len);
    }
    if (done < n) return push_iq_piece(c, rx_id, iq + 2 * (size_t)done, n - done);      // the rest, after the busy half has drained
    return CWSLG_OK;
}

int cwslg_push_iq_device(cwslg_ctx *c, int rx_id, const void *d_iq, uint32_t n)
{
    if (!c || !d_iq) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    Receiver *rx = nullptr;
    int rc = push_prologue(c, rx_id, n, &rx);
    if (rc) return rc;
    uint32_t done = 0;
    while (done < n) {
        const uint32_t ring_pos = (uint32_t)((rx->total + done) % rx->cap);
        const uint32_t m = std::min(n - done, rx->cap - ring_pos);
        HIPCHK(c, hipMemcpyAsync(rx->d_ring + ring_pos, (const float2 *)d_iq + done, (size_t)m * sizeof(float2),
                                 hipMemcpyDeviceToDevice, c->stream));
        done += m;
    }
    account_push(c, *rx, n, rx->iq_len);
    return CWSLG_OK;
}

// One block for each of n_rx receivers in ONE call: the batched form of Receiver::readIQ's per-block memcpy + inc_write_index
// (Receiver.hpp:242-249) for a host that serves thousands of streams.  Three steps like cwslg_push_iq -- (1) under the context mutex make room
// in every ring and read the write positions; (2) with no context lock, copy the callers' blocks into one pinned, host-mapped staging
// buffer; (3) under the mutex again, ONE kernel scatters the batch into the rings and the samples are accounted to the channels -- so a
// batch costs two mutex acquisitions and one launch whatever n_rx is (4096 private streams through cwslg_push_iq: 8192 and 4096 copies).
int cwslg_push_iq_many(cwslg_ctx *c, int n_rx, const int *rx_ids, const float *const *iq, uint32_t n)
{
    if (!c || n_rx <= 0 || !rx_ids || !iq) return CWSLG_ERR_ARG;
    const auto t_in = std::chrono::steady_clock::now();
    std::vector<ScatterDesc> desc((size_t)n_rx);
    std::vector<uint64_t> total0((size_t)n_rx);
    {
        std::lock_guard<std::mutex> g(c->mu);
        hipSetDevice(c->device);
        std::vector<char> seen(c->rxs.size(), 0);
        for (int k = 0; k < n_rx; ++k) {
            const int id = rx_ids[k];
            if (id < 0 || id >= (int)c->rxs.size() || !c->rxs[id].open) return fail(c, CWSLG_ERR_ARG, "cwslg_push_iq_many: bad receiver id %d", id);
            if (seen[id]) return fail(c, CWSLG_ERR_ARG, "cwslg_push_iq_many: receiver %d appears twice in one batch", id);
            seen[id] = 1;
            if (!iq[k]) return fail(c, CWSLG_ERR_ARG, "cwslg_push_iq_many: null block for receiver %d", id);
            Receiver *rx = nullptr;
            int rc = push_prologue(c, id, n, &rx);          // block-length check + room in the ring (may launch pending demodulation)
            if (rc) return rc;
            desc[k] = ScatterDesc{rx->d_ring, (unsigned)(rx->total % rx->cap), rx->cap};
            total0[k] = rx->total;
        }
    }
    // a staging buffer nobody else is filling (several threads may push disjoint batches side by side)
    BatchStage *bs = nullptr;
    std::unique_lock<std::mutex> hold;
    for (int tries = 0; tries < kBatchStages && !bs; ++tries) {
        BatchStage &cand = c->batch[(c->batch_next.fetch_add(1) + 0u) % kBatchStages];
        std::unique_lock<std::mutex> l(cand.mu, std::try_to_lock);
        if (l.owns_lock()) { bs = &cand; hold = std::move(l); }
    }
    if (!bs) { bs = &c->batch[c->batch_next.fetch_add(1) % kBatchStages]; hold = std::unique_lock<std::mutex>(bs->mu); }
    hipSetDevice(c->device);
    const size_t desc_bytes = ((size_t)n_rx * sizeof(ScatterDesc) + 255) & ~size_t(255);
    const size_t block_bytes = (size_t)n * sizeof(float2);
    const size_t need = desc_bytes + (size_t)n_rx * block_bytes;
    if (bs->busy) { HIPCHK(c, hipEventSynchronize(bs->ev)); bs->busy = false; }
    if (bs->bytes < need) {
        if (bs->h) { (void)hipHostFree(bs->h); bs->h = nullptr; bs->bytes = 0; }
        const size_t nb = (need + (1u << 20) - 1) & ~size_t((1u << 20) - 1);
        if (hipHostMalloc((void **)&bs->h, nb, hipHostMallocDefault) != hipSuccess) return fail(c, CWSLG_ERR_NOMEM, "batch staging allocation failed (%zu bytes)", nb);
        if (hipHostGetDevicePointer(&bs->h_dev, bs->h, 0) != hipSuccess) return fail(c, CWSLG_ERR_HIP, "batch staging is not device-mapped");
        bs->bytes = nb;
        if (!bs->ev) hipEventCreateWithFlags(&bs->ev, hipEventDisableTiming);
    }
    std::memcpy(bs->h, desc.data(), (size_t)n_rx * sizeof(ScatterDesc));
    for (int k = 0; k < n_rx; ++k) std::memcpy(bs->h + desc_bytes + (size_t)k * block_bytes, iq[k], block_bytes);
    {
        std::lock_guard<std::mutex> g(c->mu);
        for (int k = 0; k < n_rx; ++k) {
            const int id = rx_ids[k];
            if (id >= (int)c->rxs.size() || !c->rxs[id].open || c->rxs[id].d_ring != desc[k].ring || c->rxs[id].total != total0[k])
                return fail(c, CWSLG_ERR_ARG, "cwslg_push_iq_many: receiver %d was closed or pushed by another thread during the batch", id);
        }
        const unsigned n_pairs = n / 2;
        hipLaunchKernelGGL(scatter_blocks_kernel, dim3((n_pairs + 255) / 256, (unsigned)n_rx), dim3(256), 0, c->stream,
                           (const ScatterDesc *)bs->h_dev, (const v4f *)((const char *)bs->h_dev + desc_bytes), n_pairs);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipEventRecord(bs->ev, c->stream));
        bs->busy = true;
        for (int k = 0; k < n_rx; ++k) {
            Receiver &rx = c->rxs[rx_ids[k]];
            account_push(c, rx, n, rx.iq_len);
        }
        c->stats.h2d_bytes += (uint64_t)n_rx * block_bytes;
        c->stats.push_calls += (uint64_t)n_rx;
        c->stats.push_batches++;
        c->stats.push_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count();
    }
    return CWSLG_OK;
}

int cwslg_push_synth(cwslg_ctx *c, int rx_id, uint64_t seed, uint32_t n, uint32_t block_len,
                     const double *tones_hz, int n_tones, float amp)
{
    if (!c || n_tones < 0 || n_tones > 8 || (n_tones && !tones_hz)) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    Receiver *rx = nullptr;
    int rc = push_prologue(c, rx_id, n, &rx);
    if (rc) return rc;
    if (block_len == 0) block_len = rx->iq_len;
    if (block_len % (4 * rx->D) != 0) return fail(c, CWSLG_ERR_BLOCK, "block_len %u", block_len);
    SynthArgs a{};
    a.ring = rx->d_ring;
    a.ring_cap = rx->cap;
    a.ring_pos = rx->total % rx->cap;
    a.first_sample = rx->total;
    a.seed = seed;
    a.n = n;
    a.n_tones = n_tones;
    a.amp = amp;
    for (int t = 0; t < n_tones; ++t) {
        const double cyc = tones_hz[t] / (double)rx->fs;
        a.step[t] = (uint32_t)(int64_t)std::llround(cyc * 4294967296.0);
    }
    hipLaunchKernelGGL(synth_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, a, (const float2 *)c->d_sincos);
    HIPCHK(c, hipGetLastError());
    account_push(c, *rx, n, block_len);
    return CWSLG_OK;
}

int cwslg_ring_commit(cwslg_ctx *c, int rx_id, uint32_t n, uint32_t block_len)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    Receiver *rx = nullptr;
    int rc = push_prologue(c, rx_id, n, &rx);
    if (rc) return rc;
    if (block_len == 0) block_len = rx->iq_len;
    if (block_len % (4 * rx->D) != 0) return fail(c, CWSLG_ERR_BLOCK, "block_len %u", block_len);
    account_push(c, *rx, n, block_len);
    return CWSLG_OK;
}

int cwslg_ring_commit_all(cwslg_ctx *c, uint32_t n, uint32_t block_len)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    for (size_t id = 0; id < c->rxs.size(); ++id) {
        if (!c->rxs[id].open) continue;
        Receiver *rx = nullptr;
        int rc = push_prologue(c, (int)id, n, &rx);
        if (rc) return rc;
        const uint32_t bl = block_len ? block_len : rx->iq_len;
        if (bl % (4 * rx->D) != 0) return fail(c, CWSLG_ERR_BLOCK, "block_len %u", bl);
        account_push(c, *rx, n, bl);
    }
    return CWSLG_OK;
}

int cwslg_ring_info(cwslg_ctx *c, int rx_id, void **d_ring, uint32_t *capacity, uint64_t *total_pushed)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (rx_id < 0 || rx_id >= (int)c->rxs.size() || !c->rxs[rx_id].open) return fail(c, CWSLG_ERR_ARG, "bad receiver id");
    const Receiver &rx = c->rxs[rx_id];
    if (d_ring) *d_ring = rx.d_ring;
    if (capacity) *capacity = rx.cap;
    if (total_pushed) *total_pushed = rx.total;
    return CWSLG_OK;
}

int cwslg_channel_open(cwslg_ctx *c, int rx_id, int32_t demod_hz, int usb, const char *mode, int *ch_id)
{
    if (!c || !ch_id) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (rx_id < 0 || rx_id >= (int)c->rxs.size() || !c->rxs[rx_id].open) return fail(c, CWSLG_ERR_ARG, "bad receiver id");
    hipSetDevice(c->device);
    Receiver &rx = c->rxs[rx_id];
    const ModeInfo *mi = find_mode(mode);
    if (!mi) return fail(c, CWSLG_ERR_MODE, "Unhandled mode: %s", mode ? mode : "(null)");
    // Instance.cpp:187 : F = static_cast<float>(demodFreq) widened to double
    const double f_hz = (double)(float)demod_hz;
    int rc = check_tuning(rx.fs, kSsbBw, f_hz, usb != 0);
    if (rc) return fail(c, rc, "%s", cwslg_strerror(rc));
    Channel ch;
    ch.rx = rx_id; ch.demod_hz = demod_hz; ch.usb = usb != 0; ch.mode = mode;
    ch.group = mi->group;
    ch.wspr_scale = (ch.mode == "WSPR");
    ch.sync_ft8 = (ch.mode == "FT8");
    ch.sync_ft4 = (ch.mode == "FT4");
    ch.sync_wspr = (ch.mode == "WSPR");
    ch.sync_fst4w = (ch.mode == "FST4W-120");
    ch.frame_len = frame_length(*mi);
    ch.k = make_constants(rx.fs, kSsbBw, f_hz, usb != 0);
    ch.k_open = ch.k; ch.open_demod_hz = demod_hz; ch.open_usb = usb != 0;
    // one device allocation: 2 float frames | int16 frame | peaks | factor | tone
    const size_t fbytes = (ch.frame_len * sizeof(float) + 255) & ~size_t(255);
    const size_t ibytes = (ch.frame_len * sizeof(int16_t) + 255) & ~size_t(255);
    const size_t total = 2 * fbytes + ibytes + 256 + 256;
    HIPCHK(c, hipMalloc((void **)&ch.d_block, total));
    // from here on a failure must give the block back
#define HIPCHK_FREE(expr)                                                                          \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            (void)hipFree(ch.d_block);                                                             \
            return fail(c, CWSLG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));          \
        }                                                                                          \
    } while (0)
    ch.d_frame[0] = (float *)ch.d_block;
    ch.d_frame[1] = (float *)(ch.d_block + fbytes);
    ch.d_i16 = (int16_t *)(ch.d_block + 2 * fbytes);
    ch.d_peak = (unsigned *)(ch.d_block + 2 * fbytes + ibytes);
    ch.d_factor = (float *)(ch.d_block + 2 * fbytes + ibytes + 16);
    ch.d_tone = (float2 *)(ch.d_block + 2 * fbytes + ibytes + 256);
    HIPCHK_FREE(hipMemsetAsync(ch.d_block + 2 * fbytes + ibytes, 0, 256, c->stream));
    HIPCHK_FREE(hipMemcpyAsync(ch.d_tone, ch.k.tone.data(), ch.k.block * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    HIPCHK_FREE(sync_streams(c));   // k.tone is a temporary host vector: finish the copy now
    // phasor checkpoints: cover the first frame, which continues across the discarded partial slot
    // (up to 2 frames of blocks since creation), plus one tile of slack
    const size_t n_ckpt = ckpt_need(2 * (long long)ch.frame_len, 0) + kTile / kCkptStride;
    ch.phasor_key = std::make_tuple(rx.fs, demod_hz, usb ? 1 : 0, n_ckpt);
    PhasorTable &pt = c->phasors[ch.phasor_key];
    if (pt.refs == 0) {
        pt.d_ckpt = nullptr;
        if (hipMalloc(&pt.d_ckpt, n_ckpt * sizeof(float2)) != hipSuccess) {
            c->phasors.erase(ch.phasor_key);
            (void)hipFree(ch.d_block);
            return fail(c, CWSLG_ERR_NOMEM, "phasor table allocation failed");
        }
        pt.n_ckpt = n_ckpt;
        pt.inc = make_float2(ch.k.inc.real(), ch.k.inc.imag());
        pt.built = false;
        c->phasor_todo.push_back(ch.phasor_key);
    }
    pt.refs++;
    ch.origin_abs = (int64_t)rx.total;
    ch.pend_lo = ch.origin_abs;
    ch.open = true;
    int id = -1;
    for (size_t k = 0; k < c->chans.size(); ++k) if (!c->chans[k].open) { id = (int)k; break; }
    if (id < 0) { c->chans.push_back(ch); id = (int)c->chans.size() - 1; } else c->chans[id] = ch;
    rx.channels.push_back(id);
    *ch_id = id;
    // With the sync stage already enabled, the channel's spectra plane and result buffers are allocated NOW: left to the first emitting
    // boundary, 4096 channels' allocations made that one cwslg_slot_boundary call take 60-75 ms (round 4's paced harness) while every
    // receiver thread waited for the context lock.
    if (c->sync_cfg.enabled && (c->chans[id].sync_ft8 || c->chans[id].sync_ft4)) (void)sync_ensure_channel(c, c->chans[id]);
    return CWSLG_OK;
}

// ---- decoder= lines: the reference's grammar, field for field (CWSL_DIGI.cpp:731-837)
static bool parse_int_like_stoi(const std::string &t, long long *v)
{
    // std::stoi: optional leading whitespace and sign, at least one digit, trailing garbage ignored
    size_t i = 0;
    while (i < t.size() && std::isspace((unsigned char)t[i])) ++i;
    bool neg = false;
    if (i < t.size() && (t[i] == '+' || t[i] == '-')) { neg = t[i] == '-'; ++i; }
    if (i >= t.size() || !std::isdigit((unsigned char)t[i])) return false;
    long long acc = 0;
    while (i < t.size() && std::isdigit((unsigned char)t[i])) {
        acc = acc * 10 + (t[i] - '0');
        if (acc > 2147483648LL) return false;            // out_of_range
        ++i;
    }
    acc = neg ? -acc : acc;
    if (acc > 2147483647LL || acc < -2147483648LL) return false;
    *v = acc;
    return true;
}

int cwslg_parse_decoder_line(const char *line, double freqcal_global, cwslg_decoder_spec *out)
{
    if (!line || !out) return CWSLG_ERR_ARG;
    std::vector<std::string> f;                           // splitStringByDelim(line, ' '): getline semantics
    {
        const std::string in(line);
        size_t pos = 0;
        if (!in.empty()) {
            for (;;) {
                const size_t sp = in.find(' ', pos);
                if (sp == std::string::npos) { f.push_back(in.substr(pos)); break; }
                f.push_back(in.substr(pos, sp - pos));
                pos = sp + 1;
                if (pos == in.size()) break;              // getline does not emit a trailing empty field
            }
        }
    }
    if (f.size() < 2 || f.size() > 5) return CWSLG_ERR_ARG;               // :737-741
    long long freq = 0;
    if (!parse_int_like_stoi(f[0], &freq)) return CWSLG_ERR_ARG;          // :742 std::stoi
    const ModeInfo *mi = find_mode(f[1].c_str());
    if (!mi) return CWSLG_ERR_MODE;                                       // :797-801 unknown mode
    std::memset(out, 0, sizeof(*out));
    out->freq_hz = (uint32_t)(int)freq;
    std::snprintf(out->mode, sizeof(out->mode), "%s", mi->name);
    out->smnum = -1;
    if (f.size() >= 3) {                                                  // :815-817
        long long sm = 0;
        if (!parse_int_like_stoi(f[2], &sm)) return CWSLG_ERR_ARG;
        out->smnum = (int32_t)sm;
    }
    out->freqcal = 1.0;
    if (f.size() >= 4) {                                                  // :819-822 std::stod
        char *end = nullptr;
        const double v = std::strtod(f[3].c_str(), &end);
        if (end == f[3].c_str()) return CWSLG_ERR_ARG;
        out->freqcal = v;
    }
    if (f.size() >= 5) {                                                  // :824-830
        if (std::strcmp(mi->name, "WSPR") != 0) return CWSLG_ERR_ARG;     // "Callsigns are only supported per-decoder for WSPR decoders"
        std::snprintf(out->callsign, sizeof(out->callsign), "%s", f[4].c_str());
    }
    out->calibrated_hz = (uint32_t)((double)out->freq_hz / (freqcal_global * out->freqcal));   // :834
    out->group = mi->group;
    out->frame_len = (uint32_t)frame_length(*mi);
    out->period_s = mi->period_s;
    return CWSLG_OK;
}

int cwslg_channel_open_line(cwslg_ctx *c, int rx_id, const char *line, double freqcal_global, int *ch_id)
{
    if (!c || !ch_id) return CWSLG_ERR_ARG;
    cwslg_decoder_spec spec;
    int rc = cwslg_parse_decoder_line(line, freqcal_global, &spec);
    if (rc) return fail(c, rc, "Error parsing decoder line: %s", line ? line : "(null)");
    int32_t lo = 0;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (rx_id < 0 || rx_id >= (int)c->rxs.size() || !c->rxs[rx_id].open) return fail(c, CWSLG_ERR_ARG, "bad receiver id");
        lo = c->rxs[rx_id].lo_hz;
    }
    // Instance.cpp:183 : int32 demodFreq = calibratedSSBFreq - LO (both uint32 in the reference)
    const int32_t demod = (int32_t)(spec.calibrated_hz - (uint32_t)lo);
    return cwslg_channel_open(c, rx_id, demod, 1, spec.mode, ch_id);
}

int cwslg_channel_close(cwslg_ctx *c, int ch_id)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    hipSetDevice(c->device);
    HIPCHK(c, sync_streams(c));
    wait_fetches(c);
    Channel &ch = c->chans[ch_id];
    Receiver &rx = c->rxs[ch.rx];
    rx.channels.erase(std::remove(rx.channels.begin(), rx.channels.end(), ch_id), rx.channels.end());
    hipFree(ch.d_block);
    sync_free_channel(ch.syncbuf);
        if (ch.longbuf.d_block) { (void)hipFree(ch.longbuf.d_block); ch.longbuf = LongChannelBuffers(); }
    auto it = c->phasors.find(ch.phasor_key);
    if (it != c->phasors.end() && --it->second.refs == 0) { hipFree(it->second.d_ckpt); c->phasors.erase(it); }
    if (ch.priv.d_ckpt) (void)hipFree(ch.priv.d_ckpt);
    ch = Channel();
    return CWSLG_OK;
}

// SSBD::Tune(F, isUSB, reset = true) (SSBD.hpp:96-123): new tone and phasor step, workspace zeroed, index 0, phase (1, 0).
// Samples pushed before the call are demodulated with the old tuning; the frame keeps filling where it was.
int cwslg_channel_tune(cwslg_ctx *c, int ch_id, int32_t demod_hz, int usb)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    hipSetDevice(c->device);
    Channel &ch = c->chans[ch_id];
    Receiver &rx = c->rxs[ch.rx];
    const double f_hz = (double)(float)demod_hz;
    int rc = check_tuning(rx.fs, kSsbBw, f_hz, usb != 0);
    if (rc) return fail(c, rc, "%s", cwslg_strerror(rc));                      // the channel keeps its old tuning, as after a throw
    if ((rc = process_locked(c)) != CWSLG_OK) return rc;                        // everything already pushed: old tone
    HIPCHK(c, sync_streams(c));
    DemodConstants k = make_constants(rx.fs, kSsbBw, f_hz, usb != 0);
    const auto new_key = std::make_tuple(rx.fs, demod_hz, usb ? 1 : 0, std::get<3>(ch.phasor_key));
    if ((rc = retarget_phasor(c, ch, new_key, make_float2(k.inc.real(), k.inc.imag()))) != CWSLG_OK) return rc;
    ch.k = k;
    ch.demod_hz = demod_hz;
    ch.usb = usb != 0;
    HIPCHK(c, hipMemcpyAsync(ch.d_tone, ch.k.tone.data(), ch.k.block * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, sync_streams(c));
    ch.origin_abs = (int64_t)rx.total;                                          // workspace zeroed, phase (1,0): history restarts here
    ch.pend_lo = ch.origin_abs;
    drop_private_phasor(ch);
    return CWSLG_OK;
}

// SSBD::Tune(F, isUSB, reset) (SSBD.hpp:97-123).  reset = 0 skips :116-121: workspace, index and phase survive, i.e. the FIR
// history stays as it was mixed (old tone, old phasor sequence) and the phasor continues from its live value with the new step.
// Here: everything pushed so far is demodulated with the old tuning; the live phase is read back from the channel's checkpoint
// table (the retune point is a multiple of 4 blocks from the origin, hence a checkpoint); the channel gets a private table that
// starts at that value and a new origin at the retune point; and the next 32 outputs -- the only ones whose windows straddle the
// two tunings -- are made by demod_transition_kernel.
int cwslg_channel_tune_ex(cwslg_ctx *c, int ch_id, int32_t demod_hz, int usb, int reset)
{
    if (reset) return cwslg_channel_tune(c, ch_id, demod_hz, usb);
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    hipSetDevice(c->device);
    Channel &ch = c->chans[ch_id];
    Receiver &rx = c->rxs[ch.rx];
    const double f_hz = (double)(float)demod_hz;
    int rc = check_tuning(rx.fs, kSsbBw, f_hz, usb != 0);
    if (rc) return fail(c, rc, "%s", cwslg_strerror(rc));
    if ((rc = process_locked(c)) != CWSLG_OK) return rc;                        // everything already pushed: old tuning
    if (ch.trans_active)
        return fail(c, CWSLG_ERR_UNSUPPORTED, "a second Tune(reset = false) within 32 blocks of the previous one is not supported");
    const int64_t b0_abs = (int64_t)rx.total;
    const int64_t q0 = (b0_abs - ch.origin_abs) / (int64_t)rx.D;               // blocks since the origin: a multiple of 4
    if (q0 % kCkptStride != 0) return fail(c, CWSLG_ERR_BLOCK, "retune point is not a multiple of %d blocks from the origin", kCkptStride);
    // the live phase and the phases of the last 32 blocks, from the checkpoints q0/4 - 8 .. q0/4 of the table in use
    if ((rc = build_pending_phasors(c)) != CWSLG_OK) return rc;
    const PhasorTable &cur = ch.use_priv ? ch.priv : c->phasors[ch.phasor_key];
    const int64_t c_hi = q0 / kCkptStride, c_lo = std::max<int64_t>(0, c_hi - 8);
    if ((size_t)c_hi >= cur.n_ckpt) return fail(c, CWSLG_ERR_ARG, "checkpoint table shorter than the stream");
    float2 ck[9];
    HIPCHK(c, hipMemcpyAsync(ck, cur.d_ckpt + c_lo, (size_t)(c_hi - c_lo + 1) * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_streams(c));
    auto cmul = [](float2 a, float2 b) {                                        // cmul_exact on the host (this TU is built -ffp-contract=off)
        const float ac = a.x * b.x, bd = a.y * b.y, ad = a.x * b.y, bc = a.y * b.x;
        return make_float2(ac - bd, ad + bc);
    };
    const DemodConstants k_old = ch.k;
    const float2 inc_old = make_float2(k_old.inc.real(), k_old.inc.imag());
    auto t = std::make_shared<TransWork>();
    std::memset(t.get(), 0, sizeof(TransWork));
    t->blocks_before = (int)std::min<int64_t>(32, q0);
    for (int j = 1; j <= t->blocks_before; ++j) {                               // block q0 - j: from its checkpoint, like the kernels do
        const int64_t q = q0 - j, cq = q / kCkptStride;
        float2 p = ck[cq - c_lo];
        for (int64_t sft = cq * kCkptStride; sft < q; ++sft) p = cmul(p, inc_old);
        t->phase_old[32 - j] = p;
    }
    const float2 p0 = ck[c_hi - c_lo];                                          // the live phase (phase after q0 blocks)
    DemodConstants k = make_constants(rx.fs, kSsbBw, f_hz, usb != 0);
    const float2 inc_new = make_float2(k.inc.real(), k.inc.imag());
    {
        float2 p = p0;
        for (int j = 0; j < 32; ++j) { t->phase_new[j] = p; p = cmul(p, inc_new); }
    }
    for (uint32_t m = 0; m < k.block; ++m) {
        t->tone_old[m] = make_float2(k_old.tone[m].real(), k_old.tone[m].imag());
        t->tone_new[m] = make_float2(k.tone[m].real(), k.tone[m].imag());
    }
    t->sign = k.sign;
    // private table from the live phase; the shared table of the new tuning is what the channel returns to when its
    // demodulator is next re-created (slot boundary)
    const size_t n_ckpt = std::get<3>(ch.phasor_key);
    if ((rc = build_private_phasor(c, ch, n_ckpt, p0, inc_new)) != CWSLG_OK) return rc;
    const auto new_key = std::make_tuple(rx.fs, demod_hz, usb ? 1 : 0, n_ckpt);
    if ((rc = retarget_phasor(c, ch, new_key, inc_new)) != CWSLG_OK) return rc;
    ch.k = k;
    ch.demod_hz = demod_hz;
    ch.usb = usb != 0;
    HIPCHK(c, hipMemcpyAsync(ch.d_tone, ch.k.tone.data(), ch.k.block * sizeof(float2), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, sync_streams(c));
    ch.origin_abs = b0_abs;                                                      // block indices count from the retune point now
    ch.pend_lo = b0_abs;
    ch.use_priv = true;
    ch.trans = t;
    ch.trans_active = true;
    return CWSLG_OK;
}

int cwslg_channel_info(cwslg_ctx *c, int ch_id, uint32_t *in_size, uint32_t *out_size, uint32_t *out_rate,
                       uint32_t *delay, size_t *frame_len)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    const Channel &ch = c->chans[ch_id];
    if (in_size) *in_size = 4 * ch.k.block;     // GetInSize = 2*Fs/B
    if (out_size) *out_size = 4;                // GetOutSize
    if (out_rate) *out_rate = 2 * kSsbBw;       // GetOutRate
    if (delay) *delay = 8;                      // GetDelay = latency
    if (frame_len) *frame_len = ch.frame_len;
    return CWSLG_OK;
}

int cwslg_process(cwslg_ctx *c)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    hipSetDevice(c->device);
    if (c->process_min_outputs != 0) {              // cwslg_set_process_threshold: not worth a launch yet?
        // auto (< 0): exact mode 32 streams x 640 outputs per wave (warm-up share 32 / 640), fast mode eight 256-output tiles (history share 31 / 256)
        const uint64_t need = c->process_min_outputs > 0 ? (uint64_t)c->process_min_outputs : (c->exact ? 32u * 640u : 8u * 256u);
        bool due = false;
        for (const Receiver &rx : c->rxs) {
            if (!rx.open) continue;
            for (int id : rx.channels) {
                const Channel &ch = c->chans[id];
                if (ch.open && ch.pend_n / rx.D >= need) { due = true; break; }
            }
            if (due) break;
        }
        if (!due) { c->stats.process_deferred++; return CWSLG_OK; }
    }
    return process_locked(c, false);
}

unsigned cwslg_exact_stream_length(uint64_t total_blocks, unsigned max_blocks, unsigned cu_count, int latency)
{
    return exact5_stream_length(total_blocks, max_blocks, cu_count ? cu_count : 256u, kExact5SegCap, latency != 0);
}

int cwslg_flush(cwslg_ctx *c)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    hipSetDevice(c->device);
    return process_locked(c, false);
}

int cwslg_set_process_threshold(cwslg_ctx *c, int min_outputs)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    c->process_min_outputs = min_outputs;
    return CWSLG_OK;
}

int cwslg_slot_boundary(cwslg_ctx *c, int group, uint64_t epoch_s)
{
    if (!c || group < 0 || group >= CWSLG_NUM_GROUPS) return CWSLG_ERR_ARG;
    cwslg_rendezvous_fn fn = nullptr;
    void *user = nullptr;
    uint64_t mine = 0;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (c->rdv_pending) return fail(c, CWSLG_ERR_ARG, "cwslg_slot_boundary: a boundary opened by cwslg_slot_boundary_begin has not been ended");
        hipSetDevice(c->device);
        std::vector<int> ids;
        for (size_t k = 0; k < c->chans.size(); ++k)
            if (c->chans[k].open && c->chans[k].group == group) ids.push_back((int)k);
        int rc = boundary_locked(c, ids, epoch_s, &mine);
        if (rc) return rc;
        fn = c->rdv_fn;
        user = c->rdv_user;
        if (!fn) return CWSLG_OK;
        HIPCHK(c, sync_streams(c));        // this GPU's frames of the epoch are final ...
        drain_spans(c);
    }
    uint64_t total = mine;                                  // ... and after the rendezvous so are every other GPU's
    const int rc = fn(user, group, epoch_s, mine, &total);
    std::lock_guard<std::mutex> g(c->mu);
    if (rc < 0) return rendezvous_failed(c, rc);
    c->stats.rendezvous_calls++;
    c->stats.rendezvous_frames = total;
    c->stats.rendezvous_flags_and = c->rccl_comm ? c->rdv_flags_and.load() : c->rdv_flag.load();
    return CWSLG_OK;
}

// The same boundary in two halves, for a host that keeps the GPU busy across boundaries (bench.py with N > 1; a real-time host has
// 15 s between boundaries and uses cwslg_slot_boundary).  _begin queues the boundary's device work and returns; the host pushes and
// queues the NEXT slot's demodulation; _end then waits for the boundary's own kernels only (an event behind them, not the stream) and
// runs the rendezvous -- so the all-reduce and the host's preparation overlap the next slot's demod launch instead of idling the GPU.
// Frames and candidates of the epoch must not be fetched before _end has returned.
int cwslg_slot_boundary_begin(cwslg_ctx *c, int group, uint64_t epoch_s)
{
    if (!c || group < 0 || group >= CWSLG_NUM_GROUPS) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (c->rdv_pending) return fail(c, CWSLG_ERR_ARG, "cwslg_slot_boundary_begin: the previous boundary has not been ended");
    hipSetDevice(c->device);
    std::vector<int> ids;
    for (size_t k = 0; k < c->chans.size(); ++k)
        if (c->chans[k].open && c->chans[k].group == group) ids.push_back((int)k);
    uint64_t mine = 0;
    int rc = boundary_locked(c, ids, epoch_s, &mine);
    if (rc) return rc;
    if (!c->rdv_fn) return CWSLG_OK;
    if (!c->rdv_ready) HIPCHK(c, hipEventCreateWithFlags(&c->rdv_ready, hipEventDisableTiming));
    if (c->cand_pending) HIPCHK(c, hipStreamWaitEvent(c->stream, c->cand_done, 0));   // (lab variants that run sync work on the side stream: it belongs to this boundary)
    HIPCHK(c, hipEventRecord(c->rdv_ready, c->stream));
    c->rdv_pending = true;
    c->rdv_group = group; c->rdv_epoch = epoch_s; c->rdv_mine = mine;
    return CWSLG_OK;
}

int cwslg_slot_boundary_end(cwslg_ctx *c)
{
    if (!c) return CWSLG_ERR_ARG;
    cwslg_rendezvous_fn fn = nullptr;
    void *user = nullptr;
    int group = 0;
    uint64_t epoch = 0, mine = 0;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (!c->rdv_pending) return CWSLG_OK;
        hipSetDevice(c->device);
        c->rdv_pending = false;                                 // whatever happens below, the boundary is no longer open
        HIPCHK(c, hipEventSynchronize(c->rdv_ready));          // this GPU's frames of the epoch are final ...
        fn = c->rdv_fn; user = c->rdv_user;
        group = c->rdv_group; epoch = c->rdv_epoch; mine = c->rdv_mine;
        if (!fn) return CWSLG_OK;
    }
    uint64_t total = mine;                                      // ... and after the rendezvous so are every other GPU's
    const int rc = fn(user, group, epoch, mine, &total);
    std::lock_guard<std::mutex> g(c->mu);
    if (rc < 0) return rendezvous_failed(c, rc);
    c->stats.rendezvous_calls++;
    c->stats.rendezvous_frames = total;
    c->stats.rendezvous_flags_and = c->rccl_comm ? c->rdv_flags_and.load() : c->rdv_flag.load();
    return CWSLG_OK;
}

int cwslg_slot_boundary_channel(cwslg_ctx *c, int ch_id, uint64_t epoch_s)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    hipSetDevice(c->device);
    return boundary_locked(c, std::vector<int>{ch_id}, epoch_s);
}

int cwslg_synchronize(cwslg_ctx *c)
{
    if (!c) return CWSLG_ERR_ARG;
    // the wait itself runs WITHOUT the context mutex: receiver threads keep pushing while a clock thread waits for its boundary's kernels
    hipSetDevice(c->device);
    for (hipStream_t cs : c->copy_stream) HIPCHK(c, hipStreamSynchronize(cs));
    HIPCHK(c, sync_streams(c));
    for (hipStream_t fs : c->fetch_stream) HIPCHK(c, hipStreamSynchronize(fs));
    std::lock_guard<std::mutex> g(c->mu);
    drain_spans(c);
    return CWSLG_OK;
}

int cwslg_fetch_frame(cwslg_ctx *c, int ch_id, int16_t *dst, size_t cap, uint64_t *start_epoch, size_t *n_valid, float *factor)
{
    if (!c) return CWSLG_ERR_ARG;
    const int16_t *src = nullptr;
    const float *fac_src = nullptr;
    size_t flen = 0;
    ResultFetch rf;                                                  // ticket + lifetime, released when the copy is done or the call fails
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
        Channel &ch = c->chans[ch_id];
        if (!ch.have_frame) return CWSLG_ERR_NO_FRAME;
        hipSetDevice(c->device);
        if (dst && cap < ch.frame_len) return fail(c, CWSLG_ERR_ARG, "destination holds %zu samples, frame has %zu", cap, ch.frame_len);
        if (start_epoch) *start_epoch = ch.frame_t0;
        if (n_valid) *n_valid = ch.frame_valid;
        if (!dst && !factor) return CWSLG_OK;
        src = ch.d_i16; fac_src = ch.d_factor; flen = ch.frame_len;
        int rc = begin_result_fetch(c, ch, rf);
        if (rc) return rc;
    }
    // no context lock from here on: pushes, launches and other fetches proceed
    HIPCHK(c, hipStreamWaitEvent(rf.fs, rf.ev, 0));
    if (dst) HIPCHK(c, hipMemcpyAsync(dst, src, flen * sizeof(int16_t), hipMemcpyDeviceToHost, rf.fs));
    if (factor) HIPCHK(c, hipMemcpyAsync(factor, fac_src, sizeof(float), hipMemcpyDeviceToHost, rf.fs));
    HIPCHK(c, hipStreamSynchronize(rf.fs));
    return CWSLG_OK;
}

// Everything one slot of one channel produced, under ONE ticket: the int16 frame, the scale factor and the candidate list(s) computed from
// that frame -- what Instance.cpp:238-245 packs into one ItemToDecode (audio + startEpochTime), plus the lists a candidate-aware decoder
// would take with it.  A list is returned only if it belongs to the frame's epoch (list_kind = CWSLG_LIST_NONE otherwise: sync stage off for
// that boundary).
int cwslg_fetch_slot(cwslg_ctx *c, int ch_id, int16_t *frame, size_t cap, void *list, size_t list_bytes,
                     cwslg_ft4_sync *ft4, int max_ft4, cwslg_slot_result *out)
{
    if (!c || !out || (list_bytes > 0 && !list) || (max_ft4 > 0 && !ft4)) return CWSLG_ERR_ARG;
    std::memset(out, 0, sizeof(*out));
    const int16_t *src = nullptr; const float *fac_src = nullptr; size_t flen = 0;
    const void *lsrc = nullptr; const int *lcnt = nullptr; size_t item = 0; int lim = 0, kind = CWSLG_LIST_NONE;
    const int *nrec_src = nullptr; const char *rec_src = nullptr; int max_cand = 0;
    ResultFetch rf;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
        Channel &ch = c->chans[ch_id];
        if (!ch.have_frame) return CWSLG_ERR_NO_FRAME;
        if (frame && cap < ch.frame_len) return fail(c, CWSLG_ERR_ARG, "destination holds %zu samples, frame has %zu", cap, ch.frame_len);
        hipSetDevice(c->device);
        out->start_epoch = ch.frame_t0;
        out->n_valid = ch.frame_valid;
        src = ch.d_i16; fac_src = ch.d_factor; flen = ch.frame_len;
        if (ch.cand_t0 == ch.frame_t0 && ch.cand_t0 != 0) {
            if ((ch.sync_ft8 || ch.sync_ft4) && ch.syncbuf.d_block) {
                kind = ch.sync_ft8 ? CWSLG_LIST_FT8 : CWSLG_LIST_FT4;
                lsrc = ch.syncbuf.d_cand; lcnt = ch.syncbuf.d_ncand; item = sizeof(cwslg_candidate); max_cand = ch.syncbuf.max_cand;
                if (ch.sync_ft4 && ch.syncbuf.d_ft4c) { nrec_src = ch.syncbuf.d_nrec; rec_src = (const char *)ch.syncbuf.d_rec; }
            } else if ((ch.sync_wspr || ch.sync_fst4w) && ch.longbuf.d_block) {
                kind = ch.sync_wspr ? CWSLG_LIST_WSPR : CWSLG_LIST_FST4W;
                lsrc = ch.longbuf.w.cand; lcnt = ch.longbuf.w.ncand;
                item = ch.sync_wspr ? sizeof(cwslg_wspr_candidate) : sizeof(cwslg_fst4w_candidate);
                max_cand = ch.sync_wspr ? (int)WSPR_MAXCAND : (int)F4W_MAXCAND;
            }
            if (item) lim = (int)std::min<size_t>(list_bytes / item, (size_t)max_cand);
        }
        int rc = begin_result_fetch(c, ch, rf);
        if (rc) return rc;
    }
    int cnt = 0;
    std::vector<char> tmp((size_t)lim * item);
    HIPCHK(c, hipStreamWaitEvent(rf.fs, rf.ev, 0));
    if (frame) HIPCHK(c, hipMemcpyAsync(frame, src, flen * sizeof(int16_t), hipMemcpyDeviceToHost, rf.fs));
    HIPCHK(c, hipMemcpyAsync(&out->factor, fac_src, sizeof(float), hipMemcpyDeviceToHost, rf.fs));
    if (lcnt) HIPCHK(c, hipMemcpyAsync(&cnt, lcnt, sizeof(int), hipMemcpyDeviceToHost, rf.fs));
    if (lim > 0) HIPCHK(c, hipMemcpyAsync(tmp.data(), lsrc, tmp.size(), hipMemcpyDeviceToHost, rf.fs));
    HIPCHK(c, hipStreamSynchronize(rf.fs));
    out->list_kind = kind;
    const int total = std::max(0, std::min(cnt, max_cand));
    out->n_list = std::min(total, lim);
    if (out->n_list > 0) std::memcpy(list, tmp.data(), (size_t)out->n_list * item);
    if (nrec_src && total > 0) {                       // FT4: the coherent refinement of every candidate, candidate order then segment order
        std::vector<int> nrec((size_t)total);
        std::vector<Ft4Rec> rec((size_t)total * 3);
        HIPCHK(c, hipMemcpyAsync(nrec.data(), nrec_src, nrec.size() * sizeof(int), hipMemcpyDeviceToHost, rf.fs));
        HIPCHK(c, hipMemcpyAsync(rec.data(), rec_src, rec.size() * sizeof(Ft4Rec), hipMemcpyDeviceToHost, rf.fs));
        HIPCHK(c, hipStreamSynchronize(rf.fs));
        int k_out = 0;
        for (int k = 0; k < total; ++k)
            for (int q = 0; q < nrec[k] && q < 3; ++q) {
                if (k_out < max_ft4) std::memcpy(&ft4[k_out], &rec[(size_t)k * 3 + q], sizeof(Ft4Rec));
                ++k_out;
            }
        out->n_ft4_sync = std::min(k_out, max_ft4);
    }
    return CWSLG_OK;
}

int cwslg_write_wav(cwslg_ctx *c, int ch_id, const char *path)
{
    if (!c || !path) return CWSLG_ERR_ARG;
    size_t n = 0;
    {
        std::lock_guard<std::mutex> g(c->mu);
        if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
        if (!c->chans[ch_id].have_frame) return CWSLG_ERR_NO_FRAME;
        n = c->chans[ch_id].frame_len;
    }
    std::vector<int16_t> pcm(n);
    int rc = cwslg_fetch_frame(c, ch_id, pcm.data(), n, nullptr, nullptr, nullptr);
    if (rc) return rc;
    // WavHdr (WaveFile.hpp:19-35), little endian, packed: 46 bytes
    unsigned char h[46];
    auto u32 = [&](int off, uint32_t v) { h[off] = v & 255; h[off + 1] = (v >> 8) & 255; h[off + 2] = (v >> 16) & 255; h[off + 3] = (v >> 24) & 255; };
    auto u16 = [&](int off, uint16_t v) { h[off] = v & 255; h[off + 1] = (v >> 8) & 255; };
    const uint32_t data_len = (uint32_t)(n * sizeof(int16_t));
    std::memcpy(h + 0, "RIFF", 4); u32(4, 46 + data_len - 8);
    std::memcpy(h + 8, "WAVE", 4); std::memcpy(h + 12, "fmt ", 4); u32(16, 18);
    u16(20, 1); u16(22, 1); u32(24, 12000); u32(28, 24000); u16(32, 2); u16(34, 16); u16(36, 0);
    std::memcpy(h + 38, "data", 4); u32(42, data_len);
    FILE *f = std::fopen(path, "wb");
    if (!f) return fail(c, CWSLG_ERR_ARG, "cannot open %s", path);
    const bool ok = std::fwrite(h, 1, sizeof(h), f) == sizeof(h) && std::fwrite(pcm.data(), sizeof(int16_t), n, f) == n;
    std::fclose(f);
    return ok ? CWSLG_OK : fail(c, CWSLG_ERR_ARG, "short write to %s", path);
}

// ---- host service pieces: pure functions, no device work ----
uint64_t cwslg_slot_clock_next(int group, uint64_t after_ms) { return cwslg::host::slot_clock_next(group, after_ms); }

int cwslg_pool_sizing(const int *counts, float decoderburden, int n_decoders, int *numjt9instances, int *maxwsprdinstances)
{
    if (!counts || n_decoders < 0) return CWSLG_ERR_ARG;
    cwslg::host::pool_sizing(counts, decoderburden, n_decoders, numjt9instances, maxwsprdinstances);
    return CWSLG_OK;
}

int cwslg_find_band(const int64_t *lo_hz, const uint32_t *fs_hz, int n_bands, int64_t f_hz)
{
    if (!lo_hz || !fs_hz || n_bands < 0) return CWSLG_ERR_ARG;
    return cwslg::host::find_band(lo_hz, fs_hz, n_bands, f_hz);
}

int cwslg_parse_decode_line(const char *mode, const char *line, int64_t base_freq_hz, cwslg_spot *out)
{
    if (!mode || !line || !out) return CWSLG_ERR_ARG;
    if (!cwslg::find_mode(mode) || !std::strcmp(mode, "JS8")) return CWSLG_ERR_MODE;   // JS8's varicode frames are not restated
    return cwslg::host::parse_decode_line(mode, line, base_freq_hz, out);
}

// ---- downstream hand-off formats (layouts and rules in handoff.hpp) ----
size_t cwslg_decoder_block_bytes(int js8) { return cwslg::handoff::block_layout(js8 != 0).total; }

int cwslg_decoder_block_field(int js8, const char *name, size_t *offset, size_t *bytes)
{
    if (!name) return CWSLG_ERR_ARG;
    const cwslg::handoff::BlockLayout L = cwslg::handoff::block_layout(js8 != 0);
    const struct { const char *n; size_t off, len; } arrays[] = {
        {"ipc", L.ipc, js8 ? (size_t)0 : (size_t)12}, {"ss", L.ss, 184 * cwslg::handoff::kNsMax * 4},
        {"savg", L.savg, cwslg::handoff::kNsMax * 4}, {"sred", L.sred, 5760 * 4},
        {"d2", L.d2, cwslg::handoff::kD2Samples * 2}, {"params", L.params, L.total - L.params},
    };
    for (const auto &a : arrays)
        if (std::strcmp(a.n, name) == 0) {
            if (js8 && std::strcmp(name, "ipc") == 0) return CWSLG_ERR_ARG;
            if (offset) *offset = a.off;
            if (bytes) *bytes = a.len;
            return CWSLG_OK;
        }
    return cwslg::handoff::field_offset(L, name, offset, bytes) ? CWSLG_OK : CWSLG_ERR_ARG;
}

int cwslg_fill_decoder_block(cwslg_ctx *c, int ch_id, void *block, size_t block_bytes, int js8, int decodedepth,
                             int highest_decode_hz, uint64_t *start_epoch)
{
    if (!c || !block) return CWSLG_ERR_ARG;
    const cwslg::handoff::BlockLayout L = cwslg::handoff::block_layout(js8 != 0);
    std::unique_lock<std::mutex> lk(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    Channel &ch = c->chans[ch_id];
    if (block_bytes < L.total) return fail(c, CWSLG_ERR_ARG, "decoder block holds %zu bytes, layout needs %zu", block_bytes, L.total);
    if (!ch.have_frame) return CWSLG_ERR_NO_FRAME;
    uint8_t *blk = static_cast<uint8_t *>(block);
    if (js8) {
        if (ch.mode != "JS8") return fail(c, CWSLG_ERR_MODE, "js8 block asked for a %s channel", ch.mode.c_str());
        cwslg::handoff::fill_js8_params(L, blk, decodedepth, highest_decode_hz);
    } else if (!cwslg::handoff::fill_jt9_params(L, blk, ch.mode.c_str(), decodedepth, highest_decode_hz)) {
        return fail(c, CWSLG_ERR_MODE, "Unknown mode : %s", ch.mode.c_str());           // DecoderPool.hpp:566-570
    }
    const size_t nel = ch.frame_len < cwslg::handoff::kD2Samples ? ch.frame_len : cwslg::handoff::kD2Samples;   // :576-579
    hipSetDevice(c->device);
    if (start_epoch) *start_epoch = ch.frame_t0;
    const int16_t *src = ch.d_i16;
    ResultFetch rf;                       // the copy runs like cwslg_fetch_frame's: ticket taken here, context mutex released for its duration
    int rc = begin_result_fetch(c, ch, rf);
    if (rc) return rc;
    lk.unlock();
    HIPCHK(c, hipStreamWaitEvent(rf.fs, rf.ev, 0));
    HIPCHK(c, hipMemcpyAsync(blk + L.d2, src, nel * sizeof(int16_t), hipMemcpyDeviceToHost, rf.fs));   // :588
    HIPCHK(c, hipStreamSynchronize(rf.fs));
    return CWSLG_OK;
}

int cwslg_decoder_route(const char *mode, int transfer_shmem)
{
    if (!mode || !cwslg::find_mode(mode)) return CWSLG_ERR_MODE;
    return cwslg::handoff::uses_shared_memory(mode, transfer_shmem != 0);
}

int cwslg_decoder_command(const char *mode, int shmem_route, int numjt9threads, int decodedepth, int highest_decode_hz,
                          int wspr_cycles, float trperiod, const char *target, char *app, size_t app_cap, char *opts,
                          size_t opts_cap)
{
    if (!mode || !target || !app || !opts) return CWSLG_ERR_ARG;
    std::string a, o;
    if (!cwslg::handoff::decoder_command(mode, shmem_route != 0, numjt9threads, decodedepth, highest_decode_hz, wspr_cycles,
                                         trperiod, target, a, o))
        return CWSLG_ERR_MODE;                                                           // "Mode ... not handled"
    if (a.size() + 1 > app_cap || o.size() + 1 > opts_cap) return CWSLG_ERR_ARG;
    std::memcpy(app, a.c_str(), a.size() + 1);
    std::memcpy(opts, o.c_str(), o.size() + 1);
    return CWSLG_OK;
}

int cwslg_fetch_audio_f32(cwslg_ctx *c, int ch_id, float *dst, size_t cap, size_t *n_valid)
{
    if (!c || !dst) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    Channel &ch = c->chans[ch_id];
    if (!ch.have_frame) return CWSLG_ERR_NO_FRAME;
    if (cap < ch.frame_len) return fail(c, CWSLG_ERR_ARG, "destination holds %zu samples, frame has %zu", cap, ch.frame_len);
    hipSetDevice(c->device);
    HIPCHK(c, hipMemcpyAsync(dst, ch.d_frame[ch.frame_idx], ch.frame_valid * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, sync_streams(c));
    drain_spans(c);
    std::memset(dst + ch.frame_valid, 0, (ch.frame_len - ch.frame_valid) * sizeof(float));   // the reference's zero tail
    if (n_valid) *n_valid = ch.frame_valid;
    return CWSLG_OK;
}

int cwslg_frame_device_ptrs(cwslg_ctx *c, int ch_id, const int16_t **d_i16, const float **d_f32)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    Channel &ch = c->chans[ch_id];
    if (!ch.have_frame) return CWSLG_ERR_NO_FRAME;
    if (d_i16) *d_i16 = ch.d_i16;
    if (d_f32) *d_f32 = ch.d_frame[ch.frame_idx];
    return CWSLG_OK;
}

int cwslg_get_stats(cwslg_ctx *c, cwslg_stats *out)
{
    if (!c || !out) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    *out = c->stats;
    out->push_calls += c->push_calls_a.load(std::memory_order_relaxed);
    out->push_host_ms += 1e-6 * (double)c->push_host_ns_a.load(std::memory_order_relaxed);
    return CWSLG_OK;
}

int cwslg_reset_stats(cwslg_ctx *c)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    c->stats = cwslg_stats{};
    c->stats.rccl_world = (uint64_t)c->rccl_world;
    c->clk_sum_mhz = 0.0;
    c->clk_tail = c->clk_head;         // launches timed before the reset: neither their clock stamps ...
    c->stat_gen++;                     // ... nor their event spans count afterwards
    c->push_calls_a.store(0); c->push_host_ns_a.store(0);
    return CWSLG_OK;
}

const char *cwslg_demod_kernel_name(cwslg_ctx *c)
{
    if (!c) return "";
    std::lock_guard<std::mutex> g(c->mu);
    return c->demod_kernel_name;
}

int cwslg_set_timing(cwslg_ctx *c, int enable)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    c->timing = enable != 0;
    if (c->timing && !c->clk_h) {          // the host-mapped ring the timed exact-mode launches write their clock counters to
        hipSetDevice(c->device);
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, kClkSlots * 4 * sizeof(unsigned long long), hipHostMallocDefault) == hipSuccess &&
            hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
            c->clk_h = (unsigned long long *)h;
            c->clk_dev = (unsigned long long *)d;
        } else if (h) {
            (void)hipHostFree(h);
        }
    }
    return CWSLG_OK;
}

void *cwslg_stream(cwslg_ctx *c) { return c ? (void *)c->stream : nullptr; }

#ifdef CWSLG_STAMP
// diagnostic build only: copies the phase stamps of the last demod launch (8 x uint64 per workgroup)
int cwslg_debug_read_stamps(unsigned long long *dst, size_t n_words)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamps), n_words * 8) == hipSuccess ? 0 : -8;
}
#endif

int cwslg_channel_constants(cwslg_ctx *c, int ch_id, float *taps, float *tone_ri, float *phase_inc_ri)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    const Channel &ch = c->chans[ch_id];
    const Receiver &rx = c->rxs[ch.rx];
    if (taps) std::memcpy(taps, c->h_taps[rx.fs].data(), c->h_taps[rx.fs].size() * sizeof(float));
    if (tone_ri) std::memcpy(tone_ri, ch.k.tone.data(), ch.k.block * 2 * sizeof(float));
    if (phase_inc_ri) { phase_inc_ri[0] = ch.k.inc.real(); phase_inc_ri[1] = ch.k.inc.imag(); }
    return (int)ch.k.block;
}

int cwslg_phasor_checkpoint_stride(void) { return kCkptStride; }

int cwslg_channel_phasor_checkpoints(cwslg_ctx *c, int ch_id, float *dst_ri, size_t n, size_t *n_total)
{
    if (!c) return CWSLG_ERR_ARG;
    std::lock_guard<std::mutex> g(c->mu);
    if (ch_id < 0 || ch_id >= (int)c->chans.size() || !c->chans[ch_id].open) return fail(c, CWSLG_ERR_ARG, "bad channel id");
    hipSetDevice(c->device);
    Channel &ch = c->chans[ch_id];
    int rc = build_pending_phasors(c);
    if (rc) return rc;
    PhasorTable &pt = c->phasors[ch.phasor_key];
    if (n_total) *n_total = pt.n_ckpt;
    if (dst_ri && n) {
        const size_t m = std::min(n, pt.n_ckpt);
        HIPCHK(c, hipMemcpyAsync(dst_ri, pt.d_ckpt, m * sizeof(float2), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, sync_streams(c));
    }
    return CWSLG_OK;
}

} // extern "C"

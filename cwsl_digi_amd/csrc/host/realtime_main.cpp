// cwsl_gpu_realtime -- the ingest seam at north-star scale, paced by the wall clock, through the C ABI only (include/cwsl_gpu.h).
//
// Reference shape (Receiver.hpp:167, :209-276): ONE thread per Receiver wakes up once per block, copies BlockInSamples complex<float> into
// its ring slot and bumps the write index; every Instance of the band pulls that block.  Here the same thread makes ONE call per block,
// cwslg_push_iq(rx, block) (--mode threads), or -- for a host that serves thousands of private streams -- one thread makes one
// cwslg_push_iq_many per block period for all receivers (--mode batch).  Block k of a receiver is due at t0 + k * (block / fs) / speed
// (--speed 1 = real time); a receiver that finds its block late pushes at once (lateness is recorded, never a drop by this program:
// drops are the library's "af buffer full" count, which must stay 0).
//
// Slot boundaries are placed by SAMPLE COUNT (after --pre blocks, then every --slot-blocks blocks, on every receiver alike) so that a
// parity test can replay the identical schedule on the oracle; the pushers meet at the boundary block, one of them calls
// cwslg_slot_boundary and they go on -- the boundary's kernels, the wait for them (cwslg_synchronize, which does not hold the context
// lock while it waits) and the frame fetches (--fetch-threads, cwslg_fetch_frame on the library's fetch streams) overlap the next
// slot's pushes, exactly as DecoderPool's workers overlap the next slot in the reference.
//
// IQ: --iq FILE (raw complex64, a whole number of blocks) read cyclically, receiver r starting --iq-stride * r blocks in, so that the
// streams differ and a test can rebuild every one of them; without --iq a private xorshift noise + tone buffer (timing only).
// Output: ONE JSON line (per-boundary latencies, drops, rates, host CPU, GPU busy); --dump K --out DIR writes the last frame of K channels
// spread over the range as the reference's .wav (cwslg_write_wav) plus dump.txt "<channel> <receiver> <demod_hz> <t_start> <file>".
#include <sys/resource.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/cwsl_gpu.h"

namespace {

using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point a) { return std::chrono::duration<double, std::milli>(Clock::now() - a).count(); }

struct Options {
    int receivers = 32, channels_per_rx = 128;
    uint32_t fs = 192000, block = 2048;
    int pre_blocks = 16;           // the partial first slot (its frame is discarded, Instance.cpp:224-227)
    int slot_blocks = 1406;        // 14.997 s at 192 kHz / 2048
    int slots = 3;
    double speed = 1.0;            // 0 = unpaced
    std::string mode = "threads";  // threads | batch
    int exact = 1, sync = 1;
    double process_ms = 0;         // stream time between explicit cwslg_process calls (0: the library demodulates when a ring fills or at the boundary)
    int flush_before = 0;          // cwslg_flush this many blocks ahead of every slot boundary (a host with a slot clock knows when it is due): the boundary then
                                   // finds only these blocks pending
    int process_threshold = 0;     // cwslg_set_process_threshold: 0 every cwslg_process() launches, -1 the library's own threshold, > 0 outputs
    std::string iq_path, out_dir;
    int iq_stride = 7;
    int dump = 0, fetch_threads = 4;
    int device = 0;
    int ring_blocks = 0;           // 0 = the reference's 3 * (fs / block + 1)
};

[[noreturn]] void die(const std::string &m) { std::fprintf(stderr, "cwsl_gpu_realtime: %s\n", m.c_str()); std::exit(2); }
#define CHK(call)                                                                                                       \
    do {                                                                                                                \
        const int rc_ = (call);                                                                                         \
        if (rc_ < 0) die(std::string(#call) + ": " + cwslg_strerror(rc_) + " -- " + cwslg_last_error(g_ctx));           \
    } while (0)
cwslg_ctx *g_ctx = nullptr;

// all pushers meet here; the last one in runs `fn` while the others wait
class Meet {
public:
    explicit Meet(int n) : n_(n) {}
    template <class F> void arrive(F &&fn)
    {
        std::unique_lock<std::mutex> l(mu_);
        const uint64_t gen = gen_;
        if (++count_ == n_) { fn(); count_ = 0; ++gen_; cv_.notify_all(); }
        else cv_.wait(l, [&] { return gen_ != gen; });
    }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    uint64_t gen_ = 0;
};

int32_t channel_freq(int gch) { return -90000 + (int32_t)(((int64_t)gch * 1373) % 176000); }

}  // namespace

int main(int argc, char **argv)
{
    Options o;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) die("missing value for " + a); return argv[++i]; };
        if (a == "--receivers") o.receivers = std::atoi(val());
        else if (a == "--channels-per-rx") o.channels_per_rx = std::atoi(val());
        else if (a == "--fs") o.fs = (uint32_t)std::atol(val());
        else if (a == "--block") o.block = (uint32_t)std::atol(val());
        else if (a == "--pre") o.pre_blocks = std::atoi(val());
        else if (a == "--slot-blocks") o.slot_blocks = std::atoi(val());
        else if (a == "--slots") o.slots = std::atoi(val());
        else if (a == "--speed") o.speed = std::atof(val());
        else if (a == "--mode") o.mode = val();
        else if (a == "--exact") o.exact = std::atoi(val());
        else if (a == "--sync") o.sync = std::atoi(val());
        else if (a == "--process-ms") o.process_ms = std::atof(val());
        else if (a == "--process-threshold") o.process_threshold = std::atoi(val());
        else if (a == "--flush-before") o.flush_before = std::atoi(val());
        else if (a == "--iq") o.iq_path = val();
        else if (a == "--iq-stride") o.iq_stride = std::atoi(val());
        else if (a == "--out") o.out_dir = val();
        else if (a == "--dump") o.dump = std::atoi(val());
        else if (a == "--fetch-threads") o.fetch_threads = std::atoi(val());
        else if (a == "--device") o.device = std::atoi(val());
        else if (a == "--ring-blocks") o.ring_blocks = std::atoi(val());
        else die("unknown option " + a);
    }
    if (o.receivers < 1 || o.channels_per_rx < 1 || o.slots < 1 || o.slot_blocks < 1 || o.pre_blocks < 1) die("bad sizes");
    if (o.mode != "threads" && o.mode != "batch") die("--mode threads|batch");
    const int R = o.receivers, C = o.channels_per_rx, NCH = R * C;

    // ---- IQ source
    std::vector<float> iq;                 // interleaved re, im
    size_t iq_blocks = 0;
    if (!o.iq_path.empty()) {
        FILE *f = std::fopen(o.iq_path.c_str(), "rb");
        if (!f) die("cannot open " + o.iq_path);
        std::fseek(f, 0, SEEK_END);
        const long bytes = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        if (bytes <= 0 || bytes % (8 * (long)o.block) != 0) die("--iq: not a whole number of blocks of complex64");
        iq.resize((size_t)bytes / 4);
        if (std::fread(iq.data(), 1, (size_t)bytes, f) != (size_t)bytes) die("short read");
        std::fclose(f);
        iq_blocks = (size_t)bytes / (8 * (size_t)o.block);
    } else {
        iq_blocks = 64;
        iq.resize(iq_blocks * o.block * 2);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (size_t k = 0; k < iq.size() / 2; ++k) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            const double ph = 2.0 * M_PI * 31000.0 * (double)k / (double)o.fs;
            iq[2 * k] = (float)((int)(x & 0xFFF) - 2048) + 8000.0f * (float)std::cos(ph);
            iq[2 * k + 1] = (float)((int)((x >> 12) & 0xFFF) - 2048) + 8000.0f * (float)std::sin(ph);
        }
    }
    auto block_ptr = [&](int r, long k) -> const float * {
        const size_t b = ((size_t)r * (size_t)o.iq_stride + (size_t)k) % iq_blocks;
        return iq.data() + b * o.block * 2;
    };

    // ---- context, receivers, channels
    CHK(cwslg_create(&g_ctx, o.device));
    cwslg_ctx *ctx = g_ctx;
    CHK(cwslg_set_exact(ctx, o.exact));
    CHK(cwslg_set_process_threshold(ctx, o.process_threshold));
    if (o.sync) CHK(cwslg_enable_sync(ctx, 1, 1.5f, 200, 200, 3000));
    std::vector<int> rx_ids(R), ch_ids;
    const Clock::time_point t_setup = Clock::now();
    for (int r = 0; r < R; ++r) {
        CHK(cwslg_receiver_open(ctx, o.fs, o.block, 0, (uint32_t)o.ring_blocks, &rx_ids[r]));
        for (int c = 0; c < C; ++c) {
            int id = -1;
            CHK(cwslg_channel_open(ctx, rx_ids[r], channel_freq(r * C + c), 1, "FT8", &id));
            ch_ids.push_back(id);
        }
    }
    CHK(cwslg_synchronize(ctx));
    const double setup_ms = ms_since(t_setup);
    const size_t frame_len = 12000 * 20;

    // ---- boundary bookkeeping
    struct BoundaryRec { double call_ms, ready_ms, fetch_ms; long at_block; };
    std::vector<BoundaryRec> recs;
    std::mutex recs_mu;
    std::vector<std::thread> side_jobs;                 // one per boundary: wait for the kernels, then fetch every frame
    std::vector<std::vector<int16_t>> fetch_buf((size_t)std::max(1, o.fetch_threads), std::vector<int16_t>(frame_len));
    std::atomic<long> frames_fetched{0};
    std::mutex fetch_pool_mu;                           // one boundary's fetch at a time (they share the buffers)
    uint64_t epoch = 1000;
    auto boundary = [&](long at_block, bool emits) {
        const Clock::time_point t0 = Clock::now();
        CHK(cwslg_slot_boundary(ctx, CWSLG_GROUP_FT8, epoch));
        epoch += 15;
        const double call_ms = ms_since(t0);
        if (!emits) return;
        const size_t idx = [&] { std::lock_guard<std::mutex> g(recs_mu); recs.push_back({call_ms, 0, 0, at_block}); return recs.size() - 1; }();
        side_jobs.emplace_back([&, idx, t0] {
            CHK(cwslg_synchronize(ctx));                // the boundary's finalise + sync kernels are done: frames and candidates are final
            const double ready = ms_since(t0);
            std::lock_guard<std::mutex> pool(fetch_pool_mu);
            std::vector<std::thread> th;
            std::atomic<int> next{0};
            for (int t = 0; t < o.fetch_threads; ++t)
                th.emplace_back([&, t] {
                    for (;;) {
                        const int k = next.fetch_add(1);
                        if (k >= NCH) break;
                        uint64_t t_start = 0; size_t nv = 0; float fac = 0;
                        CHK(cwslg_fetch_frame(ctx, ch_ids[k], fetch_buf[t].data(), frame_len, &t_start, &nv, &fac));
                        frames_fetched.fetch_add(1);
                    }
                });
            for (auto &x : th) x.join();
            std::lock_guard<std::mutex> g(recs_mu);
            recs[idx].ready_ms = ready;
            recs[idx].fetch_ms = o.fetch_threads > 0 ? ms_since(t0) : 0;
        });
    };

    // ---- the paced run
    const long total_blocks = (long)o.pre_blocks + (long)o.slots * o.slot_blocks;
    const double block_s = (double)o.block / (double)o.fs;
    auto is_boundary_after = [&](long k) {              // a boundary fires after block k (0-based) has been pushed
        if (k + 1 == o.pre_blocks) return 1;            // discarded partial slot
        if (k + 1 > o.pre_blocks && (k + 1 - o.pre_blocks) % o.slot_blocks == 0) return 2;
        return 0;
    };
    std::vector<double> worst_late((size_t)R, 0.0), sum_late((size_t)R, 0.0);
    std::vector<double> worst_push((size_t)R, 0.0);
    struct rusage ru0; getrusage(RUSAGE_SELF, &ru0);
    CHK(cwslg_reset_stats(ctx));
    CHK(cwslg_set_timing(ctx, 1));
    const long proc_every = o.process_ms > 0 ? std::max<long>(1, std::lround(o.process_ms / 1e3 / block_s)) : 0;
    const Clock::time_point t_run = Clock::now();
    auto due = [&](long k) { return t_run + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(o.speed > 0 ? k * block_s / o.speed : 0.0)); };
    if (o.mode == "threads") {
        Meet meet(R);
        std::vector<std::thread> pushers;
        for (int r = 0; r < R; ++r)
            pushers.emplace_back([&, r] {
                for (long k = 0; k < total_blocks; ++k) {
                    if (o.speed > 0) std::this_thread::sleep_until(due(k));
                    const double late = o.speed > 0 ? std::max(0.0, std::chrono::duration<double, std::milli>(Clock::now() - due(k)).count()) : 0.0;
                    worst_late[r] = std::max(worst_late[r], late); sum_late[r] += late;
                    const Clock::time_point tp = Clock::now();
                    CHK(cwslg_push_iq(ctx, rx_ids[r], block_ptr(r, k), o.block));
                    worst_push[r] = std::max(worst_push[r], ms_since(tp));
                    if (r == 0 && proc_every && (k + 1) % proc_every == 0 && !is_boundary_after(k)) CHK(cwslg_process(ctx));
                    if (r == 0 && o.flush_before > 0 && is_boundary_after(k + o.flush_before) == 2 && !is_boundary_after(k)) CHK(cwslg_flush(ctx));
                    if (const int b = is_boundary_after(k)) meet.arrive([&] { boundary(k, b == 2); });
                }
            });
        for (auto &t : pushers) t.join();
    } else {
        std::vector<const float *> ptrs((size_t)R);
        for (long k = 0; k < total_blocks; ++k) {
            if (o.speed > 0) std::this_thread::sleep_until(due(k));
            const double late = o.speed > 0 ? std::max(0.0, std::chrono::duration<double, std::milli>(Clock::now() - due(k)).count()) : 0.0;
            worst_late[0] = std::max(worst_late[0], late); sum_late[0] += late;
            for (int r = 0; r < R; ++r) ptrs[r] = block_ptr(r, k);
            const Clock::time_point tp = Clock::now();
            CHK(cwslg_push_iq_many(ctx, R, rx_ids.data(), ptrs.data(), o.block));
            worst_push[0] = std::max(worst_push[0], ms_since(tp));
            if (proc_every && (k + 1) % proc_every == 0 && !is_boundary_after(k)) CHK(cwslg_process(ctx));
            if (o.flush_before > 0 && is_boundary_after(k + o.flush_before) == 2 && !is_boundary_after(k)) CHK(cwslg_flush(ctx));
            if (const int b = is_boundary_after(k)) boundary(k, b == 2);
        }
    }
    const double push_wall_ms = ms_since(t_run);
    for (auto &t : side_jobs) t.join();
    CHK(cwslg_synchronize(ctx));
    const double wall_ms = ms_since(t_run);
    struct rusage ru1; getrusage(RUSAGE_SELF, &ru1);
    auto tv = [](const timeval &a) { return (double)a.tv_sec + 1e-6 * (double)a.tv_usec; };
    const double cpu_s = tv(ru1.ru_utime) - tv(ru0.ru_utime) + tv(ru1.ru_stime) - tv(ru0.ru_stime);
    cwslg_stats st;
    CHK(cwslg_get_stats(ctx, &st));

    // ---- dump K channels spread over the range for the parity check
    if (o.dump > 0 && !o.out_dir.empty()) {
        FILE *lst = std::fopen((o.out_dir + "/dump.txt").c_str(), "w");
        if (!lst) die("cannot write to " + o.out_dir);
        for (int j = 0; j < o.dump; ++j) {
            const int k = o.dump == 1 ? 0 : (int)((int64_t)j * (NCH - 1) / (o.dump - 1));
            uint64_t t_start = 0; size_t nv = 0;
            CHK(cwslg_fetch_frame(ctx, ch_ids[k], nullptr, 0, &t_start, &nv, nullptr));
            const std::string path = o.out_dir + "/ch" + std::to_string(k) + ".wav";
            CHK(cwslg_write_wav(ctx, ch_ids[k], path.c_str()));
            if (o.sync) {
                std::vector<cwslg_candidate> cand(600);
                int n = 0;
                uint64_t t_list = 0;
                CHK(cwslg_fetch_candidates(ctx, ch_ids[k], cand.data(), 600, &n, &t_list));
                if (t_list != t_start) die("candidate list of another epoch than the frame");
                FILE *cf = std::fopen((o.out_dir + "/ch" + std::to_string(k) + ".cand").c_str(), "w");
                for (int q = 0; q < n; ++q) std::fprintf(cf, "%d %d %.9g\n", cand[q].freq_bin, cand[q].time_step, cand[q].sync);
                std::fclose(cf);
            }
            std::fprintf(lst, "%d %d %d %llu %zu %s\n", k, k / C, (int)channel_freq(k), (unsigned long long)t_start, nv, path.c_str());
        }
        std::fclose(lst);
    }

    // ---- the line
    double wl = 0, sl = 0, wp = 0;
    for (int r = 0; r < R; ++r) { wl = std::max(wl, worst_late[r]); sl += sum_late[r]; wp = std::max(wp, worst_push[r]); }
    const double n_push = (o.mode == "threads" ? (double)R : 1.0) * (double)total_blocks;
    const double stream_s = total_blocks * block_s;
    std::printf("{\"program\": \"cwsl_gpu_realtime\", \"mode\": \"%s\", \"exact\": %d, \"sync\": %d, \"receivers\": %d, \"channels_per_receiver\": %d, "
                "\"channels\": %d, \"fs_hz\": %u, \"block\": %u, \"speed\": %.3f, \"pre_blocks\": %d, \"slot_blocks\": %d, \"slots\": %d, "
                "\"stream_seconds\": %.3f, \"wall_s\": %.3f, \"push_wall_s\": %.3f, \"setup_s\": %.3f, "
                "\"blocks_pushed\": %.0f, \"blocks_dropped\": %llu, \"frames_emitted\": %llu, \"frames_discarded\": %llu, \"frames_fetched\": %ld, "
                "\"h2d_gbytes_per_s\": %.4f, \"push_calls\": %llu, \"push_batches\": %llu, \"push_host_ms_total\": %.1f, \"push_call_ms_worst\": %.3f, "
                "\"push_late_ms_worst\": %.3f, \"push_late_ms_mean\": %.4f, \"host_cpu_seconds_per_second\": %.4f, "
                "\"gpu_busy_fraction\": %.5f, \"demod_launches\": %llu, \"demod_ms\": %.2f, \"finalize_ms\": %.2f, \"sync_ms\": %.2f, "
                "\"process_every_ms\": %.1f, \"process_threshold\": %d, \"flush_before_blocks\": %d, \"process_deferred\": %llu, \"demod_redundancy\": %.4f, \"boundaries\": [",
                o.mode.c_str(), o.exact, o.sync, R, C, NCH, o.fs, o.block, o.speed, o.pre_blocks, o.slot_blocks, o.slots, stream_s, wall_ms / 1e3,
                push_wall_ms / 1e3, setup_ms / 1e3, (double)R * total_blocks, (unsigned long long)st.blocks_dropped,
                (unsigned long long)st.frames_emitted, (unsigned long long)st.frames_discarded, frames_fetched.load(),
                (double)st.h2d_bytes / 1e9 / (push_wall_ms / 1e3), (unsigned long long)st.push_calls, (unsigned long long)st.push_batches, st.push_host_ms, wp,
                wl, sl / std::max(1.0, n_push), cpu_s / (wall_ms / 1e3), (st.demod_ms + st.finalize_ms + st.sync_ms) / wall_ms,
                (unsigned long long)st.demod_launches, st.demod_ms, st.finalize_ms, st.sync_ms, o.process_ms, o.process_threshold, o.flush_before,
                (unsigned long long)st.process_deferred, st.demod_samples ? (double)st.demod_blocks_read * (o.fs / 12000.0) / (double)st.demod_samples : 0.0);
    for (size_t k = 0; k < recs.size(); ++k)
        std::printf("%s{\"after_block\": %ld, \"boundary_call_ms\": %.3f, \"frames_ready_ms\": %.3f, \"all_frames_fetched_ms\": %.3f}", k ? ", " : "",
                    recs[k].at_block, recs[k].call_ms, recs[k].ready_ms, recs[k].fetch_ms);
    std::printf("]}\n");
    cwslg_destroy(ctx);
    return 0;
}

// cwsl_gpu_skimmer -- a runnable Linux host around libcwslgpu.so (SURVEY.md 8f, rows n2 + n3): config.ini in, IQ
// from band files / stdin / UDP in place of CWSL's Win32 shared memory, slot clock from sample count or wall clock,
// and per finalised frame the decoder hand-off (12 kHz .wav, sync candidates, the jt9/wsprd command the reference
// would run).  Uses the C ABI only (include/cwsl_gpu.h), the way a CWSL_DIGI maintainer's Receiver/Instance would.
//
//   Receiver thread      Receiver.hpp:215-249   -> read block, cwslg_push_iq (one push feeds every channel of the band)
//   decoder -> band      CWSL_Utils.hpp:28-55   -> find_band over the opened receivers
//   Instance             Instance.cpp:178-288   -> cwslg_channel_open_line ... cwslg_fetch_frame / cwslg_write_wav
//   slot clocks          CWSL_DIGI.cpp:174-451  -> slot_clock_next + cwslg_slot_boundary
//   first partial slot   Instance.cpp:224-227   -> cwslg_fetch_frame returns CWSLG_ERR_NO_FRAME
//
// No GPU => cwslg_create fails and the program exits non-zero; there is no CPU path.
#include <chrono>
#include <cinttypes>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/cwsl_gpu.h"
#include "iq_source.hpp"
#include "skimmer_config.hpp"

using namespace cwslg::host;

namespace {

// The reference's Receiver: one thread per band reads blocks into a ring of 3*(Fs/iq_len + 1) slots and blocks
// ("I/Q buffer is full!") when the consumers are a lap behind (Receiver.hpp:127-153, :215-249).  Same shape here:
// the reader thread owns the source, the main thread pops whole blocks and pushes them to the GPU.
struct BlockQueue {
    std::mutex mu;
    std::condition_variable cv_put, cv_get;
    std::deque<std::vector<std::complex<float>>> q;
    size_t cap = 8;
    bool done = false, aborted = false;
    uint64_t full_events = 0;
    bool put(std::vector<std::complex<float>> &&b)            // false: the consumer has stopped
    {
        std::unique_lock<std::mutex> l(mu);
        if (q.size() >= cap) { ++full_events; cv_put.wait(l, [&] { return q.size() < cap || aborted; }); }
        if (aborted) return false;
        q.emplace_back(std::move(b));
        cv_get.notify_one();
        return true;
    }
    void abort() { std::lock_guard<std::mutex> l(mu); aborted = true; cv_put.notify_all(); }
    void finish() { std::lock_guard<std::mutex> l(mu); done = true; cv_get.notify_all(); }
    bool get(std::vector<std::complex<float>> &b)            // false: source ended and queue drained
    {
        std::unique_lock<std::mutex> l(mu);
        cv_get.wait(l, [&] { return !q.empty() || done; });
        if (q.empty()) return false;
        b = std::move(q.front());
        q.pop_front();
        cv_put.notify_one();
        return true;
    }
};

struct Rx {
    RxSpec spec; IqSource src; int id = -1; uint64_t samples = 0; bool eof = false;
    std::unique_ptr<BlockQueue> queue; std::thread reader;
};
struct Chan { int id = -1; int rx = -1; cwslg_decoder_spec spec; uint64_t frames = 0; };

int usage()
{
    std::fprintf(stderr,
        "usage: cwsl_gpu_skimmer --config config.ini --rx file=PATH|-[,header=1][,fs=N,block=N,lo=HZ] [--rx udp=PORT,fs=..]...\n"
        "         --out DIR [--start-ms UTC_MS] [--pace samples|wall] [--fast] [--sync 0|1] [--wav route|always|never]\n"
        "         [--max-seconds S] [--flush-ahead-ms MS (default 100; 0 = never)] [--device N] [--dry-run]\n"
        "         [--world N --rank R --rccl-id FILE]   one process per GPU: decoders shard by receiver (receiver k -> rank k mod N),\n"
        "                                               RCCL rendezvous at every slot boundary; rank 0 writes FILE, the others read it\n");
    return 2;
}

std::string json_escape(const std::string &s)
{
    std::string o;
    for (char ch : s) { if (ch == '"' || ch == '\\') o += '\\'; o += ch; }
    return o;
}

}  // namespace

int main(int argc, char **argv)
{
    std::string cfg_path, out_dir = ".", pace = "samples", wav_mode = "route";
    std::vector<std::string> rx_args;
    uint64_t start_ms = 0;
    bool exact = true, dry = false, have_start = false;     // exact: the library default (bit-identical frames); --fast: the fused form
    int sync = 1, device = -1, world = 1, rank = 0;
    std::string rccl_id_path;
    double max_seconds = 0;
    long flush_ahead_ms = 100;     // cwslg_flush this long before the next slot edge (0: never): the boundary then finds a few blocks pending, not seconds
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](const char *) -> const char * { return (i + 1 < argc) ? argv[++i] : nullptr; };
        const char *v = nullptr;
        if (a == "--config" && (v = need("config"))) cfg_path = v;
        else if (a == "--rx" && (v = need("rx"))) rx_args.push_back(v);
        else if (a == "--out" && (v = need("out"))) out_dir = v;
        else if (a == "--start-ms" && (v = need("start"))) { start_ms = std::strtoull(v, nullptr, 10); have_start = true; }
        else if (a == "--pace" && (v = need("pace"))) pace = v;
        else if (a == "--wav" && (v = need("wav"))) wav_mode = v;
        else if (a == "--sync" && (v = need("sync"))) sync = std::atoi(v);
        else if (a == "--device" && (v = need("device"))) device = std::atoi(v);
        else if (a == "--max-seconds" && (v = need("max"))) max_seconds = std::atof(v);
        else if (a == "--flush-ahead-ms" && (v = need("flush"))) flush_ahead_ms = std::atol(v);
        else if (a == "--world" && (v = need("world"))) world = std::atoi(v);
        else if (a == "--rank" && (v = need("rank"))) rank = std::atoi(v);
        else if (a == "--rccl-id" && (v = need("id"))) rccl_id_path = v;
        else if (a == "--exact") exact = true;              // (the default; accepted for older command lines)
        else if (a == "--fast") exact = false;
        else if (a == "--dry-run") dry = true;
        else return usage();
    }
    if (cfg_path.empty() || rx_args.empty()) return usage();
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && rccl_id_path.empty())) return usage();
    if (world > 1 && !have_start) {
        // the ranks fire the same sequence of slot boundaries (a collective) only if they count time from the same instant
        std::fprintf(stderr, "--world %d needs --start-ms: every rank must derive the slot boundaries from the same start\n", world);
        return 2;
    }

    SkimmerConfig cfg;
    if (!load_config(cfg_path.c_str(), cfg)) { std::fprintf(stderr, "config: %s\n", cfg.error.c_str()); return 1; }
    for (const std::string &n : cfg.notes) std::fprintf(stderr, "config: %s\n", n.c_str());

    std::vector<Rx> rxs(rx_args.size());
    for (size_t k = 0; k < rx_args.size(); ++k) {
        std::string err;
        if (!parse_rx_spec(rx_args[k], rxs[k].spec, err) || !rxs[k].src.open(rxs[k].spec, err)) { std::fprintf(stderr, "rx %zu: %s\n", k, err.c_str()); return 1; }
    }
    // decoder -> band (findBand); a decoder outside every band is an error in the reference too ("Unable to open CWSL shared memory")
    std::vector<int64_t> los; std::vector<uint32_t> fss;
    for (const Rx &r : rxs) { los.push_back(r.spec.lo); fss.push_back(r.spec.fs); }
    std::vector<Chan> chans;
    for (const cwslg_decoder_spec &d : cfg.decoders) {
        Chan c; c.spec = d;
        c.rx = find_band(los.data(), fss.data(), (int)rxs.size(), (int64_t)d.calibrated_hz);
        if (c.rx < 0) { std::fprintf(stderr, "decoder %u %s: no receiver covers it\n", d.freq_hz, d.mode); return 1; }
        chans.push_back(c);
    }
    // one process per GPU: a band's IQ goes to exactly one GPU (the reference creates one Receiver per
    // band, CWSL_DIGI.cpp:115-129), so receiver k and all of its decoders belong to rank k mod world
    if (world > 1) {
        std::vector<Chan> mine;
        for (const Chan &c : chans) if (c.rx % world == rank) mine.push_back(c);
        chans.swap(mine);
        std::vector<std::string> lines;
        {
            size_t k = 0;
            for (const cwslg_decoder_spec &d : cfg.decoders) {
                const int rx = find_band(los.data(), fss.data(), (int)rxs.size(), (int64_t)d.calibrated_hz);
                if (rx % world == rank) lines.push_back(cfg.decoder_lines[k]);
                ++k;
            }
        }
        cfg.decoder_lines.swap(lines);
    }
    if (!have_start) start_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();

    if (dry) {
        std::printf("{\"decoders\": %zu, \"receivers\": %zu, \"numjt9instances\": %d, \"maxwsprdinstances\": %d, \"highestdecodefreq\": %d, "
                    "\"decodedepth\": %d, \"numjt9threads\": %d, \"wsprcycles\": %d, \"transfer_shmem\": %d, \"ft_scale\": %.9g, \"wspr_scale\": %.9g, \"plan\": [",
                    chans.size(), rxs.size(), cfg.numjt9instances, cfg.maxwsprdinstances, cfg.highest_decode_hz, cfg.decodedepth,
                    cfg.numjt9threads, cfg.wspr_cycles, (int)cfg.transfer_shmem, cfg.ft_scale, cfg.wspr_scale);
        for (size_t k = 0; k < chans.size(); ++k) {
            const Chan &c = chans[k];
            std::printf("%s{\"freq_hz\": %u, \"calibrated_hz\": %u, \"mode\": \"%s\", \"rx\": %d, \"demod_hz\": %" PRId64 ", \"group\": %d, \"route\": \"%s\"}",
                        k ? ", " : "", c.spec.freq_hz, c.spec.calibrated_hz, c.spec.mode, c.rx,
                        (int64_t)c.spec.calibrated_hz - rxs[c.rx].spec.lo, c.spec.group,
                        cwslg_decoder_route(c.spec.mode, cfg.transfer_shmem) == 1 ? "shmem" : "wavefile");
        }
        std::printf("]}\n");
        return 0;
    }

    cwslg_ctx *ctx = nullptr;
    int rc = cwslg_create(&ctx, device);
    if (rc != CWSLG_OK) { std::fprintf(stderr, "cwslg_create: %s\n", cwslg_strerror(rc)); return 3; }
    auto die = [&](const char *what, int code) { std::fprintf(stderr, "%s: %s (%s)\n", what, cwslg_strerror(code), cwslg_last_error(ctx)); cwslg_destroy(ctx); std::exit(4); };
    if ((rc = cwslg_set_scale_factors(ctx, cfg.ft_scale, cfg.wspr_scale)) != CWSLG_OK) die("set_scale_factors", rc);
    if ((rc = cwslg_set_exact(ctx, exact ? 1 : 0)) != CWSLG_OK) die("set_exact", rc);
    if (sync && (rc = cwslg_enable_sync(ctx, 1, 1.5f, 200, 200, cfg.highest_decode_hz)) != CWSLG_OK) {
        // e.g. wsjtx.highestdecodefreq <= 200: no candidate search is possible; frames are still produced
        std::fprintf(stderr, "sync stage disabled: %s (%s)\n", cwslg_strerror(rc), cwslg_last_error(ctx));
        sync = 0;
    }
    if (world > 1) {
        // ncclUniqueId hand-over through a file: rank 0 creates it (written whole, then renamed), the others wait for it.  The id is
        // followed by a run tag (--start-ms, the same on every rank): a file left behind by an earlier run, or by a crash, carries
        // another tag and is ignored instead of sending ncclCommInitRank into a rendezvous nobody else attends.
        unsigned char id[CWSLG_RCCL_ID_BYTES];
        const uint64_t tag = start_ms;
        if (rank == 0) {
            std::remove(rccl_id_path.c_str());                    // whatever an earlier run left there
            if ((rc = cwslg_rccl_unique_id(id)) != CWSLG_OK) die("rccl_unique_id", rc);
            const std::string tmp = rccl_id_path + ".tmp";
            FILE *f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id || std::fwrite(&tag, 1, sizeof tag, f) != sizeof tag) {
                std::fprintf(stderr, "cannot write %s\n", tmp.c_str());
                return 1;
            }
            std::fclose(f);
            std::rename(tmp.c_str(), rccl_id_path.c_str());
        } else {
            bool got = false;
            for (int tries = 0; tries < 600 && !got; ++tries) {
                FILE *f = std::fopen(rccl_id_path.c_str(), "rb");
                if (f) {
                    uint64_t t = 0;
                    got = std::fread(id, 1, sizeof id, f) == sizeof id && std::fread(&t, 1, sizeof t, f) == sizeof t && t == tag;
                    std::fclose(f);
                }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            if (!got) { std::fprintf(stderr, "no RCCL id of this run (tag %" PRIu64 ") in %s after 60 s\n", tag, rccl_id_path.c_str()); cwslg_destroy(ctx); return 1; }
        }
        if ((rc = cwslg_rccl_init(ctx, id, rank, world)) != CWSLG_OK) die("rccl_init", rc);
    }
    for (size_t k = 0; k < rxs.size(); ++k) {
        Rx &r = rxs[k];
        if ((int)(k % (size_t)world) != rank) { r.eof = true; continue; }           // another GPU's band
        if ((rc = cwslg_receiver_open(ctx, r.spec.fs, r.spec.block, (int32_t)r.spec.lo, 0, &r.id)) != CWSLG_OK) die("receiver_open", rc);
    }
    std::set<int> groups;
    for (size_t k = 0; k < chans.size(); ++k) {
        Chan &c = chans[k];
        if ((rc = cwslg_channel_open_line(ctx, rxs[c.rx].id, cfg.decoder_lines[k].c_str(), cfg.freqcal, &c.id)) != CWSLG_OK) die("channel_open", rc);
        groups.insert(c.spec.group);
    }

    // every rank fires every group's boundaries, also for groups it owns no decoder of: the rendezvous inside
    // cwslg_slot_boundary is a collective, all ranks must make the same sequence of calls
    if (world > 1) for (const cwslg_decoder_spec &d : cfg.decoders) groups.insert(d.group);
    if (world > (int)rxs.size()) { std::fprintf(stderr, "--world %d exceeds the %zu receivers: a rank would own no band\n", world, rxs.size()); cwslg_destroy(ctx); return 1; }

    std::string log_path = out_dir + "/frames.jsonl";
    FILE *log = std::fopen(log_path.c_str(), "w");
    if (!log) { std::fprintf(stderr, "cannot write %s\n", log_path.c_str()); cwslg_destroy(ctx); return 1; }

    // next boundary instant per group
    std::vector<std::pair<int, uint64_t>> next_edge;
    for (int g : groups) next_edge.emplace_back(g, cwslg_slot_clock_next(g, start_ms));
    std::vector<int16_t> pcm;
    std::vector<cwslg_candidate> cand(600);
    std::vector<cwslg_ft4_sync> ref4(1800);
    uint64_t frames_total = 0, boundaries = 0;

    bool inputs_done = false, all_done = false;
    auto publish = [&](int group, uint64_t edge_ms) {
        const uint64_t epoch = edge_ms / 1000;                   // Instance.cpp:214: whole seconds
        if ((rc = cwslg_slot_boundary(ctx, group, epoch)) != CWSLG_OK) die("slot_boundary", rc);
        ++boundaries;
        if (world > 1) {                                         // bit 0 of every rank's flag: "my inputs are exhausted"
            cwslg_stats st;
            cwslg_get_stats(ctx, &st);
            all_done = (st.rendezvous_flags_and & 1u) != 0;
        }
        if (inputs_done) return;                                 // only here for the collective: this rank's last frames went out before
        for (Chan &c : chans) {
            if (c.spec.group != group) continue;
            pcm.resize(c.spec.frame_len);
            // the slot's results in ONE call (cwslg_fetch_slot): frame, scale factor, candidate list and FT4 refinements of the same epoch --
            // the ItemToDecode of Instance.cpp:238-245 with the lists a candidate-aware decoder would take beside it
            const bool ft = !std::strcmp(c.spec.mode, "FT8") || !std::strcmp(c.spec.mode, "FT4");
            cwslg_slot_result sr;
            rc = cwslg_fetch_slot(ctx, c.id, pcm.data(), pcm.size(), (sync && ft) ? cand.data() : nullptr,
                                  (sync && ft) ? cand.size() * sizeof(cwslg_candidate) : 0, ref4.data(), (int)ref4.size(), &sr);
            if (rc == CWSLG_ERR_NO_FRAME) continue;             // first (partial) slot: nothing to decode
            if (rc != CWSLG_OK) die("fetch_slot", rc);
            const uint64_t t0 = sr.start_epoch; const size_t nv = (size_t)sr.n_valid; const float factor = sr.factor;
            const int route = cwslg_decoder_route(c.spec.mode, cfg.transfer_shmem);
            char stem[512];
            std::snprintf(stem, sizeof stem, "%s/%" PRIu64 "_%u_%s", out_dir.c_str(), t0, c.spec.freq_hz, c.spec.mode);
            const std::string wav = std::string(stem) + ".wav";
            const bool write_wav = wav_mode == "always" || (wav_mode == "route" && route == 0);
            if (write_wav && (rc = cwslg_write_wav(ctx, c.id, wav.c_str())) != CWSLG_OK) die("write_wav", rc);
            int ncand = -1;
            if (sync && ft && (sr.list_kind == CWSLG_LIST_FT8 || sr.list_kind == CWSLG_LIST_FT4)) {
                ncand = sr.n_list;
                FILE *cf = std::fopen((std::string(stem) + ".cand").c_str(), "w");
                if (cf) {
                    for (int q = 0; q < ncand; ++q) std::fprintf(cf, "%.9g %.9g %.9g\n", cand[q].freq_hz, cand[q].dt_s, cand[q].sync);
                    std::fclose(cf);
                }
            }
            int nref = -1;
            if (sync && sr.list_kind == CWSLG_LIST_FT4) {                // FT4: coherent refinement of every candidate
                nref = sr.n_ft4_sync;
                FILE *rf = std::fopen((std::string(stem) + ".sync4").c_str(), "w");
                if (rf) {
                    for (int q = 0; q < nref; ++q)
                        std::fprintf(rf, "%.9g %.9g %.9g %.9g %d %d %d %d\n", ref4[q].f0_hz, ref4[q].f1_hz, ref4[q].dt_s, ref4[q].sync,
                                     ref4[q].ibest, ref4[q].idf, ref4[q].seg, ref4[q].cand);
                    std::fclose(rf);
                }
            }
            char app[64] = "", opts[1024] = "";
            const std::string target = route == 1 ? std::string("<shmem-key>") : wav;
            if (cwslg_decoder_command(c.spec.mode, route, cfg.numjt9threads, cfg.decodedepth, cfg.highest_decode_hz, cfg.wspr_cycles,
                                      c.spec.period_s, target.c_str(), app, sizeof app, opts, sizeof opts) != CWSLG_OK)
                app[0] = opts[0] = 0;                             // "Mode ... not handled": no command to report
            std::fprintf(log, "{\"t_start\": %" PRIu64 ", \"freq_hz\": %u, \"mode\": \"%s\", \"n_valid\": %zu, \"factor\": %.9g, \"candidates\": %d, \"ft4_refined\": %d, "
                              "\"route\": \"%s\", \"wav\": \"%s\", \"app\": \"%s\", \"opts\": \"%s\"}\n",
                         t0, c.spec.freq_hz, c.spec.mode, nv, factor, ncand, nref, route == 1 ? "shmem" : "wavefile",
                         write_wav ? json_escape(wav).c_str() : "", app, json_escape(opts).c_str());
            ++c.frames; ++frames_total;
        }
        std::fflush(log);
    };

    // due edges fire in TIME order across the groups (ties: lower group first), so that every rank makes the same sequence of
    // (collective) boundary calls however far its own clock has jumped since the last look
    auto fire_next = [&]() {
        size_t k = 0;
        for (size_t q = 1; q < next_edge.size(); ++q)
            if (next_edge[q].second < next_edge[k].second || (next_edge[q].second == next_edge[k].second && next_edge[q].first < next_edge[k].first)) k = q;
        publish(next_edge[k].first, next_edge[k].second);
        next_edge[k].second = cwslg_slot_clock_next(next_edge[k].first, next_edge[k].second);
    };
    auto fire_due = [&](uint64_t now_ms) {
        for (;;) {
            uint64_t first = ~0ull;
            for (const auto &ge : next_edge) first = std::min(first, ge.second);
            if (next_edge.empty() || first > now_ms) break;
            fire_next();
        }
    };
    for (Rx &r : rxs) {
        r.queue.reset(new BlockQueue);
        if (r.eof) { r.queue->finish(); continue; }            // not this rank's band
        r.queue->cap = 3 * (size_t)(r.spec.fs / r.spec.block + 1);             // Receiver.hpp:132
        Rx *pr = &r;
        r.reader = std::thread([pr] {
            std::vector<std::complex<float>> b;
            for (;;) {
                const uint32_t n = pr->src.read(b);
                if (n == 0) break;
                b.resize(n);
                if (!pr->queue->put(std::move(b))) break;
                b = std::vector<std::complex<float>>();
            }
            pr->queue->finish();
        });
    }
    std::fprintf(stderr, "ready: %zu receivers, %zu decoders\n", rxs.size(), chans.size());
    std::fflush(stderr);
    // main loop: always advance the receiver that is furthest behind in (virtual) time
    std::vector<std::complex<float>> blk;
    const bool wall = pace == "wall";
    uint64_t flushed_edge = 0;
    for (;;) {
        int pick = -1; double tmin = 0;
        for (size_t k = 0; k < rxs.size(); ++k) {
            if (rxs[k].eof) continue;
            const double t = (double)rxs[k].samples / rxs[k].spec.fs;
            if (pick < 0 || t < tmin) { pick = (int)k; tmin = t; }
        }
        if (pick < 0) break;
        if (max_seconds > 0 && tmin >= max_seconds) break;
        Rx &r = rxs[pick];
        if (!r.queue->get(blk)) { r.eof = true; continue; }
        const uint32_t n = (uint32_t)blk.size();
        if ((rc = cwslg_push_iq(ctx, r.id, reinterpret_cast<const float *>(blk.data()), n)) != CWSLG_OK) die("push_iq", rc);
        r.samples += n;
        // time = the slowest live receiver's sample clock (samples pacing) or the wall clock
        uint64_t now_ms;
        if (wall) now_ms = (uint64_t)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
        else {
            double t = -1;
            for (const Rx &q : rxs) { if (q.eof) continue; const double tq = (double)q.samples / q.spec.fs; if (t < 0 || tq < t) t = tq; }
            if (t < 0) t = (double)r.samples / r.spec.fs;
            now_ms = start_ms + (uint64_t)(t * 1000.0);
        }
        // a slot clock knows when the next edge is due: demodulate what is pending a little ahead of it (cwslg_flush), so that the boundary itself
        // finds only the last few blocks and the frames are final one sync stage after it (profiles/r6_realtime.json: 10 instead of 13.6 ms at 4096 channels)
        if (flush_ahead_ms > 0 && !next_edge.empty()) {
            uint64_t first = ~0ull;
            for (const auto &ge : next_edge) first = std::min(first, ge.second);
            if (first != flushed_edge && first > now_ms && first - now_ms <= (uint64_t)flush_ahead_ms) {
                if ((rc = cwslg_flush(ctx)) != CWSLG_OK) die("flush", rc);
                flushed_edge = first;
            }
        }
        fire_due(now_ms);
    }
    if (world > 1) {
        // The end of a run is collective: a rank whose bands ended (EOF, --max-seconds) goes on making the boundary calls, in the same
        // global order and without publishing anything, until EVERY rank has said so through the rendezvous flag -- the others are
        // blocked inside exactly these collectives and would otherwise wait for ever.
        inputs_done = true;
        cwslg_set_rendezvous_flag(ctx, 1);
        while (!all_done) fire_next();
    }
    for (Rx &r : rxs) r.queue->abort();
    for (Rx &r : rxs) if (r.reader.joinable()) r.reader.join();
    cwslg_synchronize(ctx);
    cwslg_stats st;
    cwslg_get_stats(ctx, &st);
    uint64_t pushed = 0, full_events = 0;
    for (Rx &r : rxs) { pushed += r.samples; full_events += r.queue->full_events; }
    std::printf("{\"pushed_samples\": %" PRIu64 ", \"reader_waits\": %" PRIu64 ", \"frames\": %" PRIu64 ", \"boundaries\": %" PRIu64 ", \"demod_samples\": %" PRIu64 ", \"blocks_dropped\": %" PRIu64 "}\n",
                pushed, full_events, frames_total, boundaries, (uint64_t)st.demod_samples, (uint64_t)st.blocks_dropped);
    std::fclose(log);
    cwslg_destroy(ctx);
    return 0;
}

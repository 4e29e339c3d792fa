// End-to-end run of the C++ shim (include/cwsl_gpu_shim.hpp) the way a CWSL_DIGI maintainer would use it:
// Receiver::readIQ -> port.push(block); slot clock -> ctx.slotBoundary(); Instance -> chan.fetch() -> ItemToDecode.
// Built and executed by tests/test_gpu_shim.py; prints "<status> <n_valid-ish> <crc32 of the int16 frame>".
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include "../include/cwsl_gpu_shim.hpp"

static uint32_t crc32(const void *data, size_t n)
{
    const uint8_t *p = static_cast<const uint8_t *>(data);
    uint32_t c = 0xFFFFFFFFu;
    for (size_t k = 0; k < n; ++k) { c ^= p[k]; for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u))); }
    return c ^ 0xFFFFFFFFu;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");            // complex64 IQ, 2048-sample blocks, written by the test
    if (!f) return 3;
    try {
        cwslgpu::Context ctx(0);
        cwslgpu::ReceiverPort rx(ctx, 192000, 2048, 28100000);
        cwslgpu::SsbChannel chan(rx, 28074000.0 - 28100000.0, true, "FT8");
        bool threw = false;
        try { cwslgpu::SsbChannel bad(rx, 97000.0, true, "FT8"); } catch (const std::invalid_argument &e) {
            threw = std::string(e.what()) == "Signal outside of band (low)";      // SSBD.hpp:101
        }
        if (!threw) return 4;
        // a second receiver with the same channel, fed block by block through Context::pushMany: its frame must equal the first one's
        cwslgpu::ReceiverPort rx2(ctx, 192000, 2048, 28100000);
        cwslgpu::SsbChannel chan2(rx2, 28074000.0 - 28100000.0, true, "FT8");
        std::vector<std::complex<float>> blk(2048);
        std::vector<std::int16_t> audio;
        std::uint64_t t0 = 0;
        ctx.slotBoundary(CWSLG_GROUP_FT8, 1000);
        if (chan.fetch(audio, t0)) return 5;          // first (partial) slot: nothing to decode (Instance.cpp:224-227)
        while (std::fread(blk.data(), sizeof(blk[0]), blk.size(), f) == blk.size()) {
            rx.push(blk.data(), 2048);
            ctx.pushMany({rx2.id()}, {blk.data()}, 2048);
        }
        // every SSBD getter (SSBD.hpp:140-154) and a Tune that fails leaves the old tuning in force (:100-103)
        if (chan.GetInRate() != 192000 || chan.GetBandwidth() != 6000 || chan.GetOutSize() != 4 || chan.GetDelay() != 8 ||
            chan.GetCarrier() != -26000.0 || !chan.IsUSB()) return 8;
        bool tune_threw = false;
        try { chan.Tune(-97000.0, true); } catch (const std::invalid_argument &e) {
            tune_threw = std::string(e.what()) == "Signal outside of band (high)" || std::string(e.what()) == "Signal outside of band (low)";
        }
        if (!tune_threw || chan.GetCarrier() != -26000.0) return 9;
        // the frame goes out the way Instance hands it to DecoderPool::push (Instance.cpp:244-245)
        int delivered = 0; std::uint64_t sink_t0 = 0; std::uint32_t sink_crc = 0; float sink_tr = 0; int sink_id = 0; std::int64_t sink_f = 0;
        std::string sink_mode, sink_cwd;
        cwslgpu::FrameSink sink([&](std::vector<std::int16_t> &&a, const std::string &mode, std::uint64_t epoch, std::int64_t base, int id,
                                    const std::string &cwd, float trperiod) {
            ++delivered; sink_t0 = epoch; sink_crc = crc32(a.data(), a.size() * 2); sink_tr = trperiod; sink_id = id; sink_f = base;
            sink_mode = mode; sink_cwd = cwd;
        });
        sink.add(chan, 28074000, 3, "/tmp/cwd3");
        if (sink.collect() != 0) return 10;           // nothing finalised yet
        ctx.slotBoundary(CWSLG_GROUP_FT8, 1015);
        if (!chan.fetch(audio, t0)) return 6;
        {
            std::vector<std::int16_t> audio2; std::uint64_t t2 = 0;
            if (!chan2.fetch(audio2, t2) || t2 != t0 || audio2 != audio) return 13;
        }
        if (sink.collect() != 1 || sink.collect() != 0) return 11;     // delivered once
        if (delivered != 1 || sink_t0 != t0 || sink_crc != crc32(audio.data(), audio.size() * 2) || sink_tr != 15.0f || sink_id != 3 ||
            sink_f != 28074000 || sink_mode != "FT8" || sink_cwd != "/tmp/cwd3") return 12;
        {   // ABI 5: FrameSink with the candidate list beside the audio: the same frame once more, the list empty (sync stage off)
            int n2 = 0; std::uint64_t e2 = 0; std::uint32_t c2 = 0; std::size_t nc2 = 99;
            cwslgpu::FrameSink sink2([&](std::vector<std::int16_t> &&a, const std::string &, std::uint64_t epoch, std::int64_t, int, const std::string &, float,
                                         std::vector<cwslg_candidate> &&cands) { ++n2; e2 = epoch; c2 = crc32(a.data(), a.size() * 2); nc2 = cands.size(); });
            sink2.add(chan, 28074000, 3, "/tmp/cwd3");
            if (sink2.collect() != 1 || sink2.collect() != 0 || n2 != 1 || e2 != t0 || c2 != crc32(audio.data(), audio.size() * 2) || nc2 != 0) return 15;
        }
        {   // ABI 5: the slot's results under one ticket -- the same frame and epoch; no list (the sync stage is off on this context)
            std::vector<std::int16_t> audio3; std::vector<cwslg_candidate> cands3; std::uint64_t t3 = 0;
            if (!chan.fetchSlot(audio3, t3, cands3) || t3 != t0 || audio3 != audio || !cands3.empty()) return 14;
        }
        std::vector<cwslg_candidate> cands; std::vector<cwslg_ft4_sync> recs;
        bool cand_threw = false;                       // sync stage not enabled on this context: the C ABI reports it, the shim throws
        try { chan.candidates(cands); } catch (const std::exception &) { cand_threw = true; }
        try { (void)chan.ft4Sync(recs); } catch (const std::exception &) { cand_threw = true; }      // (not an FT4 channel)
        (void)cand_threw;
        std::printf("OK %llu %zu %08x %zu %zu\n", (unsigned long long)t0, audio.size(), crc32(audio.data(), audio.size() * 2),
                    chan.GetInSize(), chan.GetOutRate());
    } catch (const std::exception &e) {
        std::printf("EXC %s\n", e.what());
        return 7;
    }
    std::fclose(f);
    return 0;
}

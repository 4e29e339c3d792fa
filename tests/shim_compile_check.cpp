// Compile-only check that the shim header is self-contained C++17 against the C ABI.
#include "../include/cwsl_gpu_shim.hpp"
int shim_compile_check()
{
    cwslgpu::Context ctx(0);
    cwslgpu::ReceiverPort rx(ctx, 192000, 2048, 28100000);
    cwslgpu::SsbChannel ch(rx, -26000.0, true, "FT8");
    std::vector<std::complex<float>> blk(2048);
    rx.push(blk.data(), 2048);
    ctx.slotBoundary(CWSLG_GROUP_FT8, 15);
    std::vector<std::int16_t> audio; std::uint64_t t0 = 0;
    ch.Tune(-25000.0, true);
    int delivered = 0;
    cwslgpu::FrameSink sink([&](std::vector<std::int16_t> &&, const std::string &, std::uint64_t, std::int64_t, int, const std::string &, float) { ++delivered; });
    sink.add(ch, 28074000, 7, ".");
    delivered += sink.collect();
    std::vector<cwslg_candidate> cands; std::vector<cwslg_ft4_sync> recs;
    const std::size_t getters = ch.GetInRate() + ch.GetOutRate() + ch.GetInSize() + ch.GetOutSize() + ch.GetBandwidth() + ch.GetDelay() +
                                ch.frameLength() + static_cast<std::size_t>(ch.GetCarrier()) + (ch.IsUSB() ? 1 : 0) + ch.candidates(cands) + ch.ft4Sync(recs);
    return ch.fetch(audio, t0) ? static_cast<int>(getters) + delivered : 0;
}

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
    return ch.fetch(audio, t0) ? static_cast<int>(ch.GetInSize() + ch.GetOutRate()) : 0;
}

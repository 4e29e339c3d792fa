// host_dsp.hpp -- host-side constants of the demodulator, computed once per channel / per sample rate
// and uploaded.  Product code (not the oracle): it must produce the same bits as the reference because
// the GPU kernels consume these values verbatim.
//
//   taps      : source/LowPass.hpp:16-35 (Hamming-windowed sinc in double, stored float) followed by
//               the float running-sum normalisation of source/SSBD.hpp:66-68
//   tone, inc : source/SSBD.hpp:110-114 -- std::exp(std::complex<float>) on the HOST libm, never
//               device sinf/cosf (hard part 1)
//   checks    : source/SSBD.hpp:54-55, 100-103
//   mode table: source/CWSL_DIGI.hpp:64-113 (periods), source/CWSL_DIGI_Types.hpp:83-134 (groups),
//               source/Instance.cpp:149 (frame length), :320 (the "WSPR" scale rule)
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cwsl_gpu.h"

namespace cwslg {

constexpr uint32_t kWaveSR = 12000;   // CWSL_DIGI.hpp:51
constexpr uint32_t kSsbBw = 6000;     // CWSL_DIGI.hpp:52
constexpr int kSlots = 32;            // NumWS for latency_log2 = 3

struct ModeInfo {
    const char *name;
    float period_s;
    int group;
};

inline const ModeInfo *find_mode(const char *mode)
{
    static const ModeInfo table[] = {
        {"FT8", 15.0f, CWSLG_GROUP_FT8},         {"JS8", 15.0f, CWSLG_GROUP_FT8},
        {"FT4", 7.5f, CWSLG_GROUP_FT4},          {"WSPR", 120.0f, CWSLG_GROUP_S120},
        {"Q65-30", 30.0f, CWSLG_GROUP_Q65_30},   {"JT65", 60.0f, CWSLG_GROUP_S60},
        {"FST4-60", 60.0f, CWSLG_GROUP_S60},     {"FST4-120", 120.0f, CWSLG_GROUP_S120},
        {"FST4-300", 300.0f, CWSLG_GROUP_S300},  {"FST4-900", 900.0f, CWSLG_GROUP_S900},
        {"FST4-1800", 1800.0f, CWSLG_GROUP_S1800}, {"FST4W-120", 120.0f, CWSLG_GROUP_S120},
        {"FST4W-300", 300.0f, CWSLG_GROUP_S300}, {"FST4W-900", 900.0f, CWSLG_GROUP_S900},
        {"FST4W-1800", 1800.0f, CWSLG_GROUP_S1800},
    };
    if (!mode) return nullptr;
    for (const ModeInfo &m : table)
        if (std::strcmp(mode, m.name) == 0) return &m;
    return nullptr;
}

// Instance.cpp:149 -- float period + 5, widened to double, times 12000, truncated
inline size_t frame_length(const ModeInfo &m)
{
    const float p5 = m.period_s + 5;
    return static_cast<size_t>(static_cast<double>(kWaveSR) * static_cast<double>(p5));
}

struct DemodConstants {
    uint32_t block = 0;                   // D = Fs/B/2 (input samples per output)
    uint32_t ntaps = 0;                   // 32*D
    std::vector<std::complex<float>> tone;
    std::complex<float> inc;
    float sign = 1.0f;
};

// Geometry + range checks.  Returns a CWSLG_* status.
inline int check_tuning(uint64_t fs, uint64_t bw, double f_hz, bool usb)
{
    if (bw == 0 || (fs / bw / 2) * 2 * bw != fs || fs < 4 * bw) return CWSLG_ERR_RATIO;
    const double half = static_cast<double>(fs / 2);
    if (std::fabs(f_hz) > half) return CWSLG_ERR_BAND_LOW;
    if (std::fabs(f_hz + static_cast<double>(bw) * (usb ? 1.0 : -1.0)) > half) return CWSLG_ERR_BAND_HIGH;
    return CWSLG_OK;
}

inline std::vector<float> design_taps(uint64_t fs, uint64_t bw)
{
    constexpr double pi = 3.14159265358979323846;
    const size_t order = static_cast<size_t>(8 * 2 * fs / bw);
    const double width = static_cast<double>(bw) / static_cast<double>(fs);
    std::vector<float> h(order, 0.0f);
    h[order / 2] = 1.0f;
    const double start = -1.0 * static_cast<double>(order) / 2;
    for (size_t n = 1; n < order / 2; ++n) {
        const double arg = (start + static_cast<double>(n)) * pi * width;
        const double hamming = 0.54 - 0.46 * std::cos(2.0 * pi * static_cast<double>(n) / static_cast<double>(order));
        const double val = std::sin(arg) / arg * hamming;
        h[n] = static_cast<float>(val);
        h[order - n] = static_cast<float>(val);
    }
    float total = 0.0f;
    for (float c : h) total += c;         // float accumulator, index order
    for (float &c : h) c /= total;
    return h;
}

inline DemodConstants make_constants(uint64_t fs, uint64_t bw, double f_hz, bool usb)
{
    constexpr double pi = 3.14159265358979323846;
    DemodConstants k;
    k.block = static_cast<uint32_t>(fs / bw / 2);
    k.ntaps = static_cast<uint32_t>(8 * 2 * fs / bw);
    k.sign = usb ? 1.0f : -1.0f;
    const float side = k.sign * static_cast<float>(bw);                   // float product, as in the reference
    const float delta = static_cast<float>(-2.0 * pi * (f_hz + static_cast<double>(side) / 2.0) / static_cast<double>(fs));
    k.tone.resize(k.block);
    for (uint32_t n = 0; n < k.block; ++n)
        k.tone[n] = std::exp(std::complex<float>(0.0f, delta * static_cast<float>(n)));
    k.inc = std::exp(std::complex<float>(0.0f, delta * static_cast<float>(k.block)));
    return k;
}

} // namespace cwslg

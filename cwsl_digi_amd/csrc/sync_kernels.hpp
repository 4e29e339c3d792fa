// sync_kernels.hpp -- FT8/FT4 symbol-spectra + Costas sync stage.
// Placeholder types until the stage lands; the demod/finalise path does not depend on it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cwslg {

struct SyncConfig {
    bool enabled = false;
    float syncmin = 1.5f;
    int max_cand = 200;
    int f_lo_hz = 200, f_hi_hz = 3000;
};
struct SyncShared {};
struct SyncChannelBuffers {};

inline void sync_free_channel(SyncChannelBuffers &) {}
inline void sync_free_shared(SyncShared &) {}

} // namespace cwslg

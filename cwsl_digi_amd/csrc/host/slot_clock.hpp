// slot_clock.hpp -- the slot clock as a pure function of UTC.
//
// The reference runs one polling thread per period family (CWSL_DIGI.cpp:174-451: waitForTimeQ65_30, waitForTime60,
// waitForTimeFT8, waitForTime1800/900/300/120, waitForTimeFT4) that raises every Instance's SyncPredicate when the
// wall clock enters a boundary second; the polls are 25 ms apart near a boundary (MIN_SLEEP_MS, CWSL_DIGI.hpp:60),
// so a boundary is seen 0..25 ms late.  Here the boundary INSTANTS are computed, and a driver (wall clock or sample
// count) calls cwslg_slot_boundary() at them:
//   FT8 / JS8        seconds 0, 15, 30, 45                                   (:234-262)
//   FT4              seconds 0, 15, 30, 45 and 7.4, 22.4, 37.4, 52.4           (:404-451: at seconds 7/22/37/52 the
//                    thread sleeps until 400 ms into the second, so the half slot starts at x.400, not x.500)
//   Q65-30           seconds 0, 30                                             (:174-201)
//   60 s             second 0                                                  (:203-232)
//   120/300/900/1800 second 0 of minutes divisible by 2 / 5 / 15 / 30          (:264-402)
// POSIX time has no leap seconds, so second-of-minute and minute-of-hour follow from the epoch value.
#pragma once
#include <cstdint>

#include "../../../include/cwsl_gpu.h"

namespace cwslg {
namespace host {

// offsets (ms) of the boundaries inside one repetition `cycle_ms`; returns the count
inline int clock_pattern(int group, uint64_t &cycle_ms, uint64_t offs[8])
{
    switch (group) {
    case CWSLG_GROUP_FT8:    cycle_ms = 15000; offs[0] = 0; return 1;
    case CWSLG_GROUP_FT4:    cycle_ms = 15000; offs[0] = 0; offs[1] = 7400; return 2;
    case CWSLG_GROUP_Q65_30: cycle_ms = 30000; offs[0] = 0; return 1;
    case CWSLG_GROUP_S60:    cycle_ms = 60000; offs[0] = 0; return 1;
    case CWSLG_GROUP_S120:   cycle_ms = 120000; offs[0] = 0; return 1;
    case CWSLG_GROUP_S300:   cycle_ms = 300000; offs[0] = 0; return 1;
    case CWSLG_GROUP_S900:   cycle_ms = 900000; offs[0] = 0; return 1;
    case CWSLG_GROUP_S1800:  cycle_ms = 1800000; offs[0] = 0; return 1;
    default: return 0;
    }
}

// first boundary instant (UTC ms) strictly after `after_ms`; 0 for an unknown group
inline uint64_t slot_clock_next(int group, uint64_t after_ms)
{
    uint64_t cycle = 0, offs[8];
    const int n = clock_pattern(group, cycle, offs);
    if (n == 0) return 0;
    const uint64_t base = after_ms / cycle * cycle;           // every cycle divides an hour: aligned to UTC
    for (int rep = 0; rep < 2; ++rep)
        for (int k = 0; k < n; ++k) {
            const uint64_t e = base + rep * cycle + offs[k];
            if (e > after_ms) return e;
        }
    return base + 2 * cycle;
}

}  // namespace host
}  // namespace cwslg

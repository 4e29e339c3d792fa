/*
 * host_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as cwsl_oracle.h).
 *
 * CPU restatement of the host-service rules either side of the hot path:
 *   - the slot-clock polling threads (source/CWSL_DIGI.cpp:174-451), run here in VIRTUAL time: the loops are
 *     followed statement by statement, "now" advances only through their own sleeps (MIN_SLEEP_MS = 25,
 *     MAX_SLEEP_MS = 250, CWSL_DIGI.hpp:59-60), and every preds[k]->store(true) is recorded as a fire time;
 *   - decoder-pool sizing (CWSL_DIGI.cpp:857-887);
 *   - findBand (CWSL_Utils.hpp:28-55).
 * PARITY: CWSL_DIGI.cpp needs <windows.h>/Boost and cannot be compiled here, and the reference has no tests for
 * these rules: "parity unpinned" beyond this literal restatement.
 */
#include <math.h>
#include <stdint.h>

#define MIN_SLEEP 25
#define MAX_SLEEP 250

static int sec_of(uint64_t ms) { return (int)((ms / 1000) % 60); }
static int min_of(uint64_t ms) { return (int)((ms / 60000) % 60); }
static int msec_of(uint64_t ms) { return (int)(ms % 1000); }

/* waitForTimeFT8 (:234-262), waitForTimeQ65_30 (:174-201), waitForTime60 (:203-232): one loop shape, different second sets */
static int sim_seconds(int kind, uint64_t now, uint64_t end, uint64_t *fires, int max)
{
    int go_sec = -1, go = 0, n = 0;
    while (now < end) {
        const int s = sec_of(now);
        if (go && s == go_sec) { now += MAX_SLEEP; continue; }
        if (kind == 0) go = (s == 0 || s == 15 || s == 30 || s == 45);
        else if (kind == 2) go = (s == 0 || s == 30);
        else go = (s == 0);
        if (go) { go_sec = s; if (n < max) fires[n] = now; ++n; }
        else now += MIN_SLEEP;
    }
    return n;
}

/* waitForTime120/300/900/1800 (:264-402) */
static int sim_minutes(int div, uint64_t now, uint64_t end, uint64_t *fires, int max)
{
    int go = 0, n = 0;
    while (now < end) {
        const int min_flag = (div == 2) ? ((min_of(now) & 1) == 0) : (min_of(now) % div == 0);
        const int sec_flag = sec_of(now) == 0;
        if (min_flag && sec_flag && go) { now += MAX_SLEEP; continue; }
        go = min_flag && sec_flag;
        if (go) { if (n < max) fires[n] = now; ++n; }
        else if (min_flag || (!min_flag && sec_of(now) <= 55)) now += MAX_SLEEP;
        else now += MIN_SLEEP;
    }
    return n;
}

/* waitForTimeFT4 (:404-451) */
static int sim_ft4(uint64_t now, uint64_t end, uint64_t *fires, int max)
{
    int go_sec = -1, go = 0, n = 0;
    while (now < end) {
        const int s = sec_of(now);
        if (go && s == go_sec) { now += MAX_SLEEP; continue; }
        go = 0;
        switch (s) {
        case 0: case 15: case 30: case 45:
            go = 1;
            break;
        case 7: case 22: case 37: case 52:
            while (msec_of(now) < 300 && s == sec_of(now)) now += (uint64_t)(400 - msec_of(now));
            go = 1;
            break;
        default:
            break;
        }
        if (go) { go_sec = sec_of(now); if (n < max) fires[n] = now; ++n; }
        else now += MIN_SLEEP;
    }
    return n;
}

/* group: the CWSLG_GROUP_* numbering (FT8 0, FT4 1, Q65-30 2, 60 s 3, 120 s 4, 300 s 5, 900 s 6, 1800 s 7).
 * Returns the number of fires in [start_ms, end_ms); the first `max` fire times are stored. */
int orc_clock_sim(int group, uint64_t start_ms, uint64_t end_ms, uint64_t *fires, int max)
{
    switch (group) {
    case 0: return sim_seconds(0, start_ms, end_ms, fires, max);
    case 1: return sim_ft4(start_ms, end_ms, fires, max);
    case 2: return sim_seconds(2, start_ms, end_ms, fires, max);
    case 3: return sim_seconds(3, start_ms, end_ms, fires, max);
    case 4: return sim_minutes(2, start_ms, end_ms, fires, max);
    case 5: return sim_minutes(5, start_ms, end_ms, fires, max);
    case 6: return sim_minutes(15, start_ms, end_ms, fires, max);
    case 7: return sim_minutes(30, start_ms, end_ms, fires, max);
    default: return -1;
    }
}

/* CWSL_DIGI.cpp:857-887.  counts: FT4, FT8, Q65-30, JS8, WSPR, JT65, FST4W, FST4 */
void orc_pool_sizing(const int *counts, float decoderburden, int n_decoders, int *numjt9, int *maxwsprd)
{
    const int numFT4 = counts[0], numFT8 = counts[1], numQ65 = counts[2], numJS8 = counts[3], numWSPR = counts[4],
              numJT65 = counts[5], numFST4W = counts[6], numFST4 = counts[7];
    const float nd1 = (float)(numFT4 + numFT8 + numQ65 + numJS8) * (1.0f / 5.0f);
    const float nd2 = (float)(numWSPR) * (1.0f / 3.0f);
    const float nd3 = (float)(numJT65) * (1.0f / 3.0f);
    const float nd4 = (float)(numFST4W) * (1.0f / 3.0f);
    const float nd5 = (float)(numFST4) * (1.0f / 3.0f);
    const float nInstf = (nd1 + nd2 + nd3 + nd4 + nd5) * decoderburden;
    const int nj = (int)roundf(nInstf + 0.55f);
    int nw = (int)round((double)nj * ((double)numWSPR / (double)n_decoders));
    if (nw < 1 && numWSPR) nw = 1;
    *numjt9 = nj;
    *maxwsprd = nw;
}

/* CWSL_Utils.hpp:28-55 */
int orc_find_band(const int64_t *lo, const uint32_t *fs, int n, int64_t f)
{
    for (int b = 0; b < n; ++b)
        if (((int)fs[b] > 0) && (f >= lo[b] - (int)fs[b] / 2) && (f <= lo[b] + (int)fs[b] / 2)) return b;
    return -1;
}

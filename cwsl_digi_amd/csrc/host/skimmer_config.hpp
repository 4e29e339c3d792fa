// skimmer_config.hpp -- config.ini as the reference reads it, for the keys that reach the hot path and its
// hand-off.  The file format is boost::program_options' INI dialect: [section] headers,
// key=value, '#' comments, repeated keys for multitoken options (decoders.decoder).
//   decoder lines          CWSL_DIGI.cpp:731-837   (cwslg_parse_decoder_line)
//   pool sizing            CWSL_DIGI.cpp:846-887
//   highestdecodefreq      :889-896   (clamped to SSB_BW)
//   decodedepth            :939-950   (clamped to 1..3)
//   scale factors          :952-978   (0 < f <= 1, else fatal)
//   wsprcycles             :995-1008  (100..10000, else fatal)
//   numjt9threads          :1010-1021 (clamped to 1..9)
//   transfermethod         default shmem
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../../include/cwsl_gpu.h"

namespace cwslg {
namespace host {

// decoder counts in the order the C ABI documents: FT4, FT8, Q65-30, JS8, WSPR, JT65, FST4W, FST4
enum { CNT_FT4, CNT_FT8, CNT_Q65, CNT_JS8, CNT_WSPR, CNT_JT65, CNT_FST4W, CNT_FST4, CNT_N };

inline int count_slot(const char *mode)
{
    const std::string m = mode;
    if (m == "FT4") return CNT_FT4;
    if (m == "FT8") return CNT_FT8;
    if (m == "Q65-30") return CNT_Q65;
    if (m == "JS8") return CNT_JS8;
    if (m == "WSPR") return CNT_WSPR;
    if (m == "JT65") return CNT_JT65;
    if (m.compare(0, 6, "FST4W-") == 0) return CNT_FST4W;
    if (m.compare(0, 5, "FST4-") == 0) return CNT_FST4;
    return -1;
}

// CWSL_DIGI.cpp:857-887 -- float arithmetic as written there
inline void pool_sizing(const int counts[CNT_N], float decoderburden, int n_decoders, int *numjt9, int *maxwsprd)
{
    const float nd1 = static_cast<float>(counts[CNT_FT4] + counts[CNT_FT8] + counts[CNT_Q65] + counts[CNT_JS8]) * (1.0f / 5.0f);
    const float nd2 = static_cast<float>(counts[CNT_WSPR]) * (1.0f / 3.0f);
    const float nd3 = static_cast<float>(counts[CNT_JT65]) * (1.0f / 3.0f);
    const float nd4 = static_cast<float>(counts[CNT_FST4W]) * (1.0f / 3.0f);
    const float nd5 = static_cast<float>(counts[CNT_FST4]) * (1.0f / 3.0f);
    const float inst = (nd1 + nd2 + nd3 + nd4 + nd5) * decoderburden;
    const int nj = static_cast<int>(std::round(inst + 0.55f));
    int nw = 0;
    if (n_decoders > 0)
        nw = static_cast<int>(std::round(static_cast<double>(nj) * (static_cast<double>(counts[CNT_WSPR]) / static_cast<double>(n_decoders))));
    if (nw < 1 && counts[CNT_WSPR]) nw = 1;
    if (numjt9) *numjt9 = nj;
    if (maxwsprd) *maxwsprd = nw;
}

// CWSL_Utils.hpp:28-55 findBand: the first receiver whose band [L0 - Fs/2, L0 + Fs/2] holds f
inline int find_band(const int64_t *lo_hz, const uint32_t *fs, int n, int64_t f_hz)
{
    for (int b = 0; b < n; ++b)
        if (fs[b] > 0 && f_hz >= lo_hz[b] - (int64_t)(fs[b] / 2) && f_hz <= lo_hz[b] + (int64_t)(fs[b] / 2)) return b;
    return -1;
}

struct SkimmerConfig {
    std::vector<std::string> decoder_lines;
    std::vector<cwslg_decoder_spec> decoders;
    double freqcal = 1.0;
    int sharedmem = -1;
    float ft_scale = 0.90f, wspr_scale = 0.20f;
    int highest_decode_hz = 3000, decodedepth = 3, wspr_cycles = 3000, numjt9threads = 3;
    int numjt9instances = 0, maxwsprdinstances = 0;
    float decoderburden = 1.0f;
    bool transfer_shmem = true, keepwav = false;
    std::string temppath, binpath, js8_binpath, callsign, grid;
    std::vector<std::string> notes;          // the reference's "setting to N" corrections, in order
    std::string error;                       // non-empty: the reference would exit with this message
};

inline std::string trim(const std::string &s)
{
    size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}

inline bool parse_bool(const std::string &v)
{
    return v == "true" || v == "1" || v == "yes" || v == "on";
}

inline bool load_config(const char *path, SkimmerConfig &cfg)
{
    FILE *f = std::fopen(path, "r");
    if (!f) { cfg.error = std::string("cannot open ") + path; return false; }
    std::string section;
    std::multimap<std::string, std::string> kv;
    std::vector<std::string> order;
    char buf[4096];
    while (std::fgets(buf, sizeof buf, f)) {
        std::string line = buf;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        line = trim(line);
        if (line.empty()) continue;
        if (line.front() == '[' && line.back() == ']') { section = trim(line.substr(1, line.size() - 2)); continue; }
        const size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        const std::string key = (section.empty() ? "" : section + ".") + trim(line.substr(0, eq));
        const std::string val = trim(line.substr(eq + 1));
        kv.emplace(key, val);
        if (key == "decoders.decoder") cfg.decoder_lines.push_back(val);
    }
    std::fclose(f);
    auto has = [&](const char *k) { return kv.count(k) != 0; };
    auto get = [&](const char *k) { return kv.find(k)->second; };
    if (has("radio.freqcalibration")) cfg.freqcal = std::atof(get("radio.freqcalibration").c_str());
    if (has("radio.sharedmem")) cfg.sharedmem = std::atoi(get("radio.sharedmem").c_str());
    if (has("operator.callsign")) cfg.callsign = get("operator.callsign");
    if (has("operator.gridsquare")) cfg.grid = get("operator.gridsquare");
    if (has("wsjtx.temppath")) cfg.temppath = get("wsjtx.temppath");
    if (has("wsjtx.binpath")) cfg.binpath = get("wsjtx.binpath");
    if (has("js8call.binpath")) cfg.js8_binpath = get("js8call.binpath");
    if (has("wsjtx.keepwav")) cfg.keepwav = parse_bool(get("wsjtx.keepwav"));
    if (has("wsjtx.transfermethod")) cfg.transfer_shmem = get("wsjtx.transfermethod") == "shmem";
    if (cfg.decoder_lines.empty()) { cfg.error = "decoders.decoder input is required but was not specified!"; return false; }   // :839
    int counts[CNT_N] = {0};
    for (const std::string &l : cfg.decoder_lines) {
        cwslg_decoder_spec d;
        const int rc = cwslg_parse_decoder_line(l.c_str(), cfg.freqcal, &d);
        if (rc != CWSLG_OK) { cfg.error = "bad decoder line: " + l; return false; }
        cfg.decoders.push_back(d);
        const int s = count_slot(d.mode);
        if (s >= 0) counts[s]++;
    }
    if (has("wsjtx.decoderburden")) cfg.decoderburden = (float)std::atof(get("wsjtx.decoderburden").c_str());
    if (has("wsjtx.numjt9instances")) {
        cfg.numjt9instances = std::atoi(get("wsjtx.numjt9instances").c_str());
        if (cfg.numjt9instances < 1) { cfg.error = "wsjtx.numjt9instances must be >= 1"; return false; }
    }
    int nj = 0, nw = 0;
    pool_sizing(counts, cfg.decoderburden, (int)cfg.decoders.size(), &nj, &nw);
    if (!cfg.numjt9instances) cfg.numjt9instances = nj;
    else {                                                        // :879 uses the configured instance count
        nw = (int)std::round((double)cfg.numjt9instances * ((double)counts[CNT_WSPR] / (double)cfg.decoders.size()));
        if (nw < 1 && counts[CNT_WSPR]) nw = 1;
    }
    if (has("wsjtx.maxwsprdinstances")) {
        cfg.maxwsprdinstances = std::atoi(get("wsjtx.maxwsprdinstances").c_str());
        if (cfg.maxwsprdinstances < 1) { cfg.error = "wsjtx.maxwsprdinstances must be >= 1"; return false; }
    } else cfg.maxwsprdinstances = nw;
    if (has("wsjtx.highestdecodefreq")) cfg.highest_decode_hz = std::atoi(get("wsjtx.highestdecodefreq").c_str());
    if (cfg.highest_decode_hz > 6000) cfg.highest_decode_hz = 6000;                                   // :893-895 (SSB_BW)
    if (has("wsjtx.decodedepth")) {
        cfg.decodedepth = std::atoi(get("wsjtx.decodedepth").c_str());
        if (cfg.decodedepth > 3) { cfg.notes.push_back("wsjtx.decodedepth is too high, setting to 3"); cfg.decodedepth = 3; }
        else if (cfg.decodedepth < 1) { cfg.notes.push_back("wsjtx.decodedepth is too small, setting to 1"); cfg.decodedepth = 1; }
    }
    if (has("wsjtx.ftaudioscalefactor")) {
        cfg.ft_scale = (float)std::atof(get("wsjtx.ftaudioscalefactor").c_str());
        if (cfg.ft_scale > 1.0f) { cfg.error = "ftaudioscalefactor must be <= 1.0"; return false; }
        if (cfg.ft_scale <= 0.0f) { cfg.error = "ftaudioscalefactor must be > 0"; return false; }
    }
    if (has("wsjtx.wspraudioscalefactor")) {
        cfg.wspr_scale = (float)std::atof(get("wsjtx.wspraudioscalefactor").c_str());
        if (cfg.wspr_scale > 1.0f) { cfg.error = "wsjtx.wspraudioscalefactor must be <= 1.0"; return false; }
        if (cfg.wspr_scale <= 0.0f) { cfg.error = "wsjtx.wspraudioscalefactor must be > 0"; return false; }
    }
    if (has("wsjtx.wsprcycles")) {
        cfg.wspr_cycles = std::atoi(get("wsjtx.wsprcycles").c_str());
        if (cfg.wspr_cycles > 10000) { cfg.error = "wsjtx.wsprcycles must be <= 10000"; return false; }
        if (cfg.wspr_cycles < 100) { cfg.error = "wsjtx.wsprcycles must be >= 100"; return false; }
    }
    if (has("wsjtx.numjt9threads")) {
        cfg.numjt9threads = std::atoi(get("wsjtx.numjt9threads").c_str());
        if (cfg.numjt9threads > 9) { cfg.notes.push_back("wsjtx.numjt9threads is too high, setting to 9"); cfg.numjt9threads = 9; }
        else if (cfg.numjt9threads < 1) { cfg.notes.push_back("wsjtx.numjt9threads is too small, setting to 1"); cfg.numjt9threads = 1; }
    }
    return true;
}

}  // namespace host
}  // namespace cwslg

// handoff.hpp -- the downstream data formats of the hot path: the shared-memory
// block a stock jt9 / js8 decoder maps, and the command line it is started with.  Host-only byte work; the only
// device traffic is the D2H of the finalised int16 frame straight into the block's d2 array.
//
//   jt9 block : DecoderPool.hpp:58-108  (dec_data_t,     "MUST be kept in sync with lib/jt9com.f90")
//   js8 block : DecoderPool.hpp:110-171 (dec_data_js8_t)
//   field fill: DecoderPool.hpp:451-577 (jt9), :760-791 (js8)
//   commands  : DecoderPool.hpp:634-659 (shared memory), :1007-1046 (wave file), dispatch :379-395
//
// The layouts are described as field tables and the offsets are derived with the C ABI's rules (x86-64, MSVC and
// SysV agree here: int/float 4-byte aligned, bool/char 1 byte) -- the oracle states the same layouts as C structs,
// and tests/test_handoff.py requires both to agree byte for byte.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

namespace cwslg {
namespace handoff {

constexpr size_t kNsMax = 6827;                    // DecoderPool.hpp:44
constexpr size_t kD2Samples = 30 * 60 * 12000;     // NTMAX * RX_SAMPLE_RATE (:45-46)

enum Kind : uint8_t { I32, F32, B8, CH };
struct Field { const char *name; Kind kind; uint32_t count; };

// params of dec_data_t, in declaration order
constexpr Field kJt9Params[] = {
    {"nutc", I32, 1},      {"ndiskdat", B8, 1},   {"ntrperiod", I32, 1},  {"nQSOProgress", I32, 1}, {"nfqso", I32, 1},
    {"nftx", I32, 1},      {"newdat", B8, 1},     {"npts8", I32, 1},      {"nfa", I32, 1},          {"nfSplit", I32, 1},
    {"nfb", I32, 1},       {"ntol", I32, 1},      {"kin", I32, 1},        {"nzhsym", I32, 1},       {"nsubmode", I32, 1},
    {"nagain", B8, 1},     {"ndepth", I32, 1},    {"lft8apon", B8, 1},    {"lapcqonly", B8, 1},     {"ljt65apon", B8, 1},
    {"napwid", I32, 1},    {"ntxmode", I32, 1},   {"nmode", I32, 1},      {"minw", I32, 1},         {"nclearave", B8, 1},
    {"minSync", I32, 1},   {"emedelay", F32, 1},  {"dttol", F32, 1},      {"nlist", I32, 1},        {"listutc", I32, 10},
    {"n2pass", I32, 1},    {"nranera", I32, 1},   {"naggressive", I32, 1}, {"nrobust", B8, 1},      {"nexp_decode", I32, 1},
    {"datetime", CH, 20},  {"mycall", CH, 12},    {"mygrid", CH, 6},      {"hiscall", CH, 12},      {"hisgrid", CH, 6},
};
// params of dec_data_js8_t
constexpr Field kJs8Params[] = {
    {"nutc", I32, 1},      {"ndiskdat", B8, 1},   {"ntrperiod", I32, 1},  {"nQSOProgress", I32, 1}, {"nfqso", I32, 1},
    {"nftx", I32, 1},      {"newdat", B8, 1},     {"npts8", I32, 1},      {"nfa", I32, 1},          {"nfb", I32, 1},
    {"ntol", I32, 1},      {"syncStats", B8, 1},  {"kin", I32, 1},        {"kposA", I32, 1},        {"kposB", I32, 1},
    {"kposC", I32, 1},     {"kposE", I32, 1},     {"kposI", I32, 1},      {"kszA", I32, 1},         {"kszB", I32, 1},
    {"kszC", I32, 1},      {"kszE", I32, 1},      {"kszI", I32, 1},       {"nzhsym", I32, 1},       {"nsubmode", I32, 1},
    {"nsubmodes", I32, 1}, {"nagain", B8, 1},     {"ndepth", I32, 1},     {"lft8apon", B8, 1},      {"lapcqonly", B8, 1},
    {"ljt65apon", B8, 1},  {"napwid", I32, 1},    {"ntxmode", I32, 1},    {"nmode", I32, 1},        {"minw", I32, 1},
    {"nclearave", B8, 1},  {"minSync", I32, 1},   {"emedelay", F32, 1},   {"dttol", F32, 1},        {"nlist", I32, 1},
    {"listutc", I32, 10},  {"n2pass", I32, 1},    {"nranera", I32, 1},    {"naggressive", I32, 1},  {"nrobust", B8, 1},
    {"nexp_decode", I32, 1}, {"datetime", CH, 20}, {"mycall", CH, 12},    {"mygrid", CH, 6},        {"hiscall", CH, 12},
    {"hisgrid", CH, 6},    {"ndebug", I32, 1},
};

constexpr size_t field_align(Kind k) { return (k == I32 || k == F32) ? 4 : 1; }
constexpr size_t field_size(const Field &f) { return ((f.kind == I32 || f.kind == F32) ? 4 : 1) * (size_t)f.count; }
constexpr size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct BlockLayout {
    bool js8;
    size_t ipc, ss, savg, sred, d2, params, total;
    const Field *fields;
    size_t n_fields;
};

template <size_t N>
constexpr size_t params_size(const Field (&f)[N])
{
    size_t off = 0;
    for (size_t k = 0; k < N; ++k) off = align_up(off, field_align(f[k].kind)) + field_size(f[k]);
    return align_up(off, 4);
}

inline BlockLayout block_layout(bool js8)
{
    BlockLayout L{};
    L.js8 = js8;
    size_t off = 0;
    L.ipc = off;  if (!js8) off += 3 * 4;                          // int ipc[3] exists only in the jt9 block
    L.ss = off;   off += 184 * kNsMax * 4;
    L.savg = off; off += kNsMax * 4;
    L.sred = off; off += 5760 * 4;
    L.d2 = off;   off += kD2Samples * 2;
    L.params = align_up(off, 4);
    L.fields = js8 ? kJs8Params : kJt9Params;
    L.n_fields = js8 ? sizeof(kJs8Params) / sizeof(Field) : sizeof(kJt9Params) / sizeof(Field);
    L.total = L.params + (js8 ? params_size(kJs8Params) : params_size(kJt9Params));
    return L;
}

// offset of params.<name> inside the block; returns false for an unknown name
inline bool field_offset(const BlockLayout &L, const char *name, size_t *offset, size_t *bytes)
{
    size_t off = 0;
    for (size_t k = 0; k < L.n_fields; ++k) {
        const Field &f = L.fields[k];
        off = align_up(off, field_align(f.kind));
        if (std::strcmp(f.name, name) == 0) {
            if (offset) *offset = L.params + off;
            if (bytes) *bytes = field_size(f);
            return true;
        }
        off += field_size(f);
    }
    return false;
}

class BlockWriter {
public:
    BlockWriter(const BlockLayout &l, uint8_t *base) : L(l), p(base) {}
    void i32(const char *name, int32_t v) { put(name, &v, 4); }
    void f32(const char *name, float v) { put(name, &v, 4); }
    void b8(const char *name, bool v) { const uint8_t b = v ? 1 : 0; put(name, &b, 1); }
    int32_t get_i32(const char *name) const
    {
        size_t off = 0; int32_t v = 0;
        if (field_offset(L, name, &off, nullptr)) std::memcpy(&v, p + off, 4);
        return v;
    }
private:
    void put(const char *name, const void *src, size_t n)
    {
        size_t off = 0;
        if (field_offset(L, name, &off, nullptr)) std::memcpy(p + off, src, n);
    }
    const BlockLayout &L;
    uint8_t *p;
};

inline bool is_fst4(const char *m) { return std::strncmp(m, "FST4-", 5) == 0; }     // CWSL_DIGI.hpp:151-153
inline bool is_fst4w(const char *m) { return std::strncmp(m, "FST4W-", 6) == 0; }   // CWSL_DIGI.hpp:155-157

// DecoderPool.hpp:379-395 -- which route an item takes when transfermethod=shmem; 1 = shared memory, 0 = wave file
inline int uses_shared_memory(const char *mode, bool transfer_shmem)
{
    if (!transfer_shmem || std::strcmp(mode, "WSPR") == 0) return 0;
    if (std::strcmp(mode, "JS8") == 0 || is_fst4(mode) || is_fst4w(mode)) return 0;
    return 1;
}

// Everything of the jt9 block except d2 (DecoderPool.hpp:451-577).  Returns false for a mode that route rejects
// ("Unknown mode", :566-570).
inline bool fill_jt9_params(const BlockLayout &L, uint8_t *blk, const char *mode, int depth, int highest_hz)
{
    std::memset(blk, 0, L.total);
    BlockWriter w(L, blk);
    w.i32("nfa", 0); w.i32("nfb", highest_hz); w.i32("ndepth", depth);
    w.b8("newdat", true); w.f32("dttol", 4.0f);
    struct Row { const char *mode; int nzhsym, nmode, ntxmode, napwid, ntrperiod, ntol, nfa, nfb, nfqso, nexp; bool ft8ap, depth1; };
    static const Row rows[] = {
        //  mode        nzhsym nmode ntx napwid period ntol  nfa   nfb nfqso nexp  ap     depth1
        {"FT8",             0,    8,  0,   50,    15,   0,   -1,   -1,    0,   0, true,  false},
        {"FT4",             0,    5,  0,   80,     7,   0,   -1,   -1,    0,   0, false, false},   // (int)7.5
        {"Q65-30",        196,   66, 66,    0,    30,   0,   -1,   -1,    0,   0, false, false},
        {"JT65",          174,   65, 65,    0,    60,   0,   -1,   -1,    0,   0, false, false},
        {"FST4-60",       187,  240,  0,    0,    60, 100,  900, 1100,    0,   0, false, true},
        {"FST4-120",      387,  240,  0,    0,   120, 100,  900, 1100,    0,   0, false, true},
        {"FST4-300",     1003,  240,  0,    0,   300, 100,  700, 1100,    0,   0, false, true},
        {"FST4-900",     3107,  240,  0,    0,   900, 100,  900, 1100,    0,   0, false, true},
        {"FST4-1800",    6232,  240,  0,    0,  1800, 100,  900, 1100,    0,   0, false, true},
        {"FST4W-120",     387,  241,  0,    0,   120, 100,   -1,   -1, 1500, 768, false, false},
        {"FST4W-300",    1003,  241,  0,    0,   300, 100,   -1,   -1, 1500, 768, false, false},
        {"FST4W-900",    3107,  241,  0,    0,   900, 100,   -1,   -1, 1500, 768, false, false},
        {"FST4W-1800",   6232,  241,  0,    0,  1800, 100,   -1,   -1, 1500, 768, false, false},
    };
    const Row *r = nullptr;
    for (const Row &q : rows) if (std::strcmp(q.mode, mode) == 0) r = &q;
    if (!r) return false;
    w.b8("lft8apon", r->ft8ap);
    w.i32("nzhsym", r->nzhsym); w.i32("nmode", r->nmode); w.i32("ntxmode", r->ntxmode);
    w.i32("napwid", r->napwid); w.i32("ntrperiod", r->ntrperiod); w.i32("ntol", r->ntol);
    if (r->depth1) w.i32("ndepth", 1);
    if (r->nfa >= 0) { w.i32("nfa", r->nfa); w.i32("nfb", r->nfb); }
    w.i32("nfqso", r->nfqso); w.i32("nexp_decode", r->nexp);
    const int32_t ipc[3] = {r->nzhsym, 1, -1};                        // nzhsym, istart, idone (:572-574)
    std::memcpy(blk + L.ipc, ipc, sizeof(ipc));
    return true;
}

// DecoderPool.hpp:760-791
inline void fill_js8_params(const BlockLayout &L, uint8_t *blk, int depth, int highest_hz)
{
    std::memset(blk, 0, L.total);
    BlockWriter w(L, blk);
    w.i32("nfa", 0); w.i32("nfb", highest_hz); w.i32("ndepth", depth);
    w.b8("newdat", true); w.f32("dttol", 4.0f);
    w.i32("ntrperiod", -1); w.i32("nsubmode", -1); w.i32("n2pass", 1); w.i32("npts8", 50 * 6912 / 16);
    w.i32("kszA", (int32_t)kD2Samples - 1); w.i32("kposA", 0); w.i32("nsubmodes", 1);
    w.i32("nmode", 8); w.i32("napwid", 50);
}

// The decoder's program name and argument string, character for character as the reference concatenates them
// (leading blank and the double blank before "-s" included).  `target` = shared-memory key or wave-file name.
inline bool decoder_command(const char *mode, bool shmem_route, int threads, int depth, int highest_hz, int wspr_cycles,
                            float trperiod, const char *target, std::string &app, std::string &opts)
{
    const std::string m = " -m " + std::to_string(threads) + " ";
    const std::string M = mode;
    std::string op = " ";
    app = "jt9.exe";
    if (shmem_route) {                                                  // :634-659
        if (M == "FT8") op += "-8" + m;
        else if (M == "FT4") op += "-5" + m;
        else if (M == "Q65-30") op += "-3" + m + "-p 30 -H " + std::to_string(highest_hz) + " ";
        else if (M == "JT65") op += "-6" + m;
        else if (is_fst4w(mode)) op += "-W" + m;
        else if (is_fst4(mode)) op += "-7" + m;
        else return false;
        opts = op + " -s " + target;
        return true;
    }
    const std::string d = "-d " + std::to_string(depth) + " ";
    const std::string p = " -p " + std::to_string((int)trperiod);
    if (M == "FT8") op += "-8" + m + d + "-w 1 -H " + std::to_string(highest_hz) + " ";      // :1007-1046
    else if (M == "FT4") op += "-5" + m + d + "-w 1 -H " + std::to_string(highest_hz) + " ";
    else if (M == "Q65-30") op += "-3 -p 30 -H " + std::to_string(highest_hz) + " ";
    else if (M == "WSPR") { app = "wsprd.exe"; op += "-C " + std::to_string(wspr_cycles) + " -o 5 -d "; }
    else if (M == "JT65") op += "-6 " + d;
    else if (is_fst4w(mode)) op += "-W" + p + m + d + "-L 1400 -H 1600 -F 200 ";
    else if (is_fst4(mode)) op += "-7" + p + m;
    else if (M == "JS8") { app = "js8.exe"; op += "-8" + m; }
    else return false;
    opts = op + target;
    return true;
}

}  // namespace handoff
}  // namespace cwslg

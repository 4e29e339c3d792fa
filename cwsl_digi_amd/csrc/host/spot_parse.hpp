// spot_parse.hpp -- decoder stdout -> spot record for FT8/FT4/FST4/FST4W/WSPR: the text stage downstream of the
// decoder hand-off.  Pure host text logic, no device work.
//   line grammar     OutputHandler.cpp:505-621  parseOutputFT4FT8: "HHMMSS snr  dt freq ~  message", fixed columns
//   message rules    OutputHandler.cpp:924-1128 handleMessageUniversal: which token is the transmitting station's call,
//                    whether a grid locator follows
//   call / locator   OutputHandler.cpp:788-874 (parseCall, isCallPacked, checkCall), :889-922 (isSOTAMATMessage),
//                    HamUtils.hpp:26-43 (isValidLocator), StringUtils.hpp:11-28 (trim)
// The reporter back ends (PSKReporter / RBN / WSPRNet) and the ignore list are out of scope: the result says what
// reporter->handle() would have been called with.
#pragma once
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/cwsl_gpu.h"

namespace cwslg {
namespace host {

inline void trim_ws(std::string &s)
{
    size_t a = 0;
    while (a < s.size() && std::isspace((unsigned char)s[a])) ++a;
    size_t b = s.size();
    while (b > a && std::isspace((unsigned char)s[b - 1])) --b;
    s = s.substr(a, b - a);
}
inline bool locator_ok(const std::string &l)
{
    return l.size() == 4 && std::isalpha((unsigned char)l[0]) && std::isalpha((unsigned char)l[1]) &&
           std::isdigit((unsigned char)l[2]) && std::isdigit((unsigned char)l[3]);
}
inline bool call_packed(const std::string &c) { return !c.empty() && c.front() == '<' && c.back() == '>' && c.length() >= 5; }
inline void unpack_call(std::string &c) { if (call_packed(c)) c = c.substr(1, c.length() - 2); }
inline bool call_ok(const std::string &c)
{
    if (c.size() < 3) return false;
    size_t letters = 0;
    for (char ch : c) if (std::isalpha((unsigned char)ch)) ++letters;
    if (letters == c.size() || letters == 0) return false;               // "QRP", "POTA"; or no letter at all
    if (c.find_first_of(" .+-?;=~") != std::string::npos) return false;
    if (c.length() == 4 && std::isalpha((unsigned char)c[0]) && std::isalpha((unsigned char)c[1]) &&
        std::isdigit((unsigned char)c[2]) && std::isdigit((unsigned char)c[3])) return false;    // a grid, or RR73
    return true;
}
inline bool sotamat(const std::string &prefix, const std::string &call_sfx)
{
    if (prefix.length() + call_sfx.length() + 1 != 13) return false;
    static const char *known[] = {"S", "SM", "STM", "STMT", "SOTAM", "SOTAMT", "SOTAMAT"};
    bool hit = false;
    for (const char *k : known) hit = hit || prefix == k;
    if (!hit) return false;
    const size_t pos = call_sfx.find_first_of('/');
    if (pos == std::string::npos) return false;
    const std::string sfx = call_sfx.substr(pos + 1);
    if (sfx.length() < 2 || sfx.length() > 4) return false;
    return call_ok(call_sfx.substr(0, pos));
}

// handleMessageUniversal: true and (call[, loc]) = what the reporter is handed; false = "Message not handled"
inline bool message_to_spot(std::string msg, std::string &call, std::string &loc, bool &has_loc)
{
    call.clear(); loc.clear(); has_loc = false;
    trim_ws(msg);
    static const char *chop[] = {"?", "a1", "a2", "q0", "q1", "q2", "q3", "q4", "q5"};
    for (const char *c : chop) {
        const size_t q = msg.find(c);
        if (q != std::string::npos) { msg = msg.substr(0, q); trim_ws(msg); }
    }
    if (msg.length() < 6) return false;
    std::vector<size_t> sp;
    for (size_t k = 0; k < msg.length(); ++k) if (msg[k] == ' ') sp.push_back(k);
    const size_t n = sp.size();
    if (n == 0) return false;
    const bool cq = msg[0] == 'C' && msg[1] == 'Q';
    auto rest = [&](size_t from) { return from < msg.length() ? msg.substr(from) : std::string(); };
    if (cq && n == 1 && msg[2] == ' ') {                                          // CQ CALL
        std::string c = msg.substr(3); unpack_call(c);
        if (call_ok(c)) { call = c; return true; }
    } else if (cq && n == 2) {                                                    // CQ CALL GRID | CQ CALL x | CQ x CALL
        std::string c = msg.substr(sp[0] + 1, sp[1] - sp[0] - 1); unpack_call(c);
        const std::string l = rest(sp[1] + 1);
        if (call_ok(c)) { call = c; if (locator_ok(l)) { loc = l; has_loc = true; } return true; }
        std::string c2 = l; unpack_call(c2);
        if (call_ok(c2)) { call = c2; return true; }
    } else if (cq && n == 3) {                                                    // CQ x CALL GRID
        std::string c = msg.substr(sp[1] + 1, sp[2] - sp[1] - 1); unpack_call(c);
        const std::string l = rest(sp[2] + 1);
        if (call_ok(c) && locator_ok(l)) { call = c; loc = l; has_loc = true; return true; }
    } else if (!cq) {
        if (n == 1) {                                                             // <...> CALL ; SOTAmat
            std::string c = rest(sp[0] + 1); unpack_call(c);
            const std::string dx = msg.substr(0, sp[0]);
            if (call_packed(dx) && call_ok(c)) { call = c; return true; }
            if (sotamat(dx, c)) { call = c; return true; }
        } else if (n == 2) {                                                      // CALL CALL rpt|73|GRID
            std::string c = msg.substr(sp[0] + 1, sp[1] - sp[0] - 1); unpack_call(c);
            if (call_ok(c)) { call = c; return true; }
        } else if (n == 3) {
            std::string c = msg.substr(sp[0] + 1, sp[1] - sp[0] - 1); unpack_call(c);
            if (sp[2] - sp[1] == 2 && msg[sp[2] - 1] == 'R') {                    // CALL CALL R GRID
                const std::string l = rest(sp[2] + 1);
                if (call_ok(c) && locator_ok(l)) { call = c; loc = l; has_loc = true; return true; }
            } else if (sp[2] - sp[1] == 4) {                                      // CALL CALL RST STATE|SERIAL
                if (call_ok(c)) { call = c; return true; }
            }
        }
    }
    return false;
}

// splitStringByDelim(line, ' ', true) (StringUtils.hpp:48-68)
inline std::vector<std::string> tokens_of(const std::string &line)
{
    std::vector<std::string> v;
    size_t k = 0;
    while (k < line.size()) {
        while (k < line.size() && line[k] == ' ') ++k;
        size_t e = k;
        while (e < line.size() && line[e] != ' ') ++e;
        if (e > k) v.push_back(line.substr(k, e - k));
        k = e;
    }
    return v;
}
inline bool num_ok(const std::string &s) { char *e = nullptr; std::strtod(s.c_str(), &e); return e != s.c_str(); }

// parseOutputWSPR / parseOutputFST4W / parseOutputFST4 for ONE line (OutputHandler.cpp:314-402, 152-240, 243-312)
inline int parse_token_line(const std::string &mode, std::string line, int64_t base_freq_hz, cwslg_spot *out)
{
    trim_ws(line);
    const std::vector<std::string> tok = tokens_of(line);
    const bool wspr = mode == "WSPR", fst4w = mode.compare(0, 6, "FST4W-") == 0;
    if (wspr) { if (tok.size() != 8) return CWSLG_SPOT_SKIP; }
    else {
        if (tok.size() < (fst4w ? 8u : 4u) || line.size() <= 22) return CWSLG_SPOT_SKIP;   // the reference indexes without checking
        if (line[18] != ' ' || line[19] != '`' || line[20] != ' ' || line[21] != ' ') return CWSLG_SPOT_SKIP;
    }
    if (!num_ok(tok[1]) || !num_ok(tok[2]) || !num_ok(tok[3])) return CWSLG_SPOT_SKIP;
    out->snr_db = (int32_t)std::strtol(tok[1].c_str(), nullptr, 10);
    out->dt_s = std::strtof(tok[2].c_str(), nullptr);
    const double f = std::strtod(tok[3].c_str(), nullptr);
    out->freq_hz = (uint32_t)((double)base_freq_hz + (wspr ? f * 1000000.0 : f));
    if (wspr || fst4w) {
        std::string call = tok[5];
        if (wspr) unpack_call(call);                         // "WSPR callsigns may be packed"; FST4W's are taken as they come
        if (wspr) { if (!num_ok(tok[4]) || !num_ok(tok[7])) return CWSLG_SPOT_SKIP; out->drift = (int32_t)std::strtol(tok[4].c_str(), nullptr, 10); }
        else if (!num_ok(tok[7])) return CWSLG_SPOT_SKIP;
        out->dbm = (int32_t)std::strtol(tok[7].c_str(), nullptr, 10);
        if (!call_ok(call)) return CWSLG_SPOT_UNHANDLED;
        std::strncpy(out->call, call.c_str(), sizeof(out->call) - 1);
        std::strncpy(out->locator, tok[6].c_str(), sizeof(out->locator) - 1);
        out->has_locator = 1;
        return CWSLG_SPOT_OK;
    }
    std::string msg = line.substr(22);                       // FST4: free text through the FT8 message rules
    trim_ws(msg);
    std::strncpy(out->message, msg.c_str(), sizeof(out->message) - 1);
    std::string call, loc; bool has_loc = false;
    if (!message_to_spot(msg, call, loc, has_loc)) return CWSLG_SPOT_UNHANDLED;
    std::strncpy(out->call, call.c_str(), sizeof(out->call) - 1);
    if (has_loc) std::strncpy(out->locator, loc.c_str(), sizeof(out->locator) - 1);
    out->has_locator = has_loc ? 1 : 0;
    return CWSLG_SPOT_OK;
}

// parseOutputFT4FT8 for ONE line.  Returns a CWSLG_SPOT_* status; on CWSLG_SPOT_OK / _UNHANDLED the numeric fields are set.
inline int parse_decode_line(const char *mode, const char *line_in, int64_t base_freq_hz, cwslg_spot *out)
{
    std::memset(out, 0, sizeof *out);
    std::string line = line_in ? line_in : "";
    const bool jt65 = std::strcmp(mode, "JT65") == 0;
    const bool col = jt65 || !std::strcmp(mode, "FT8") || !std::strcmp(mode, "FT4") || !std::strcmp(mode, "Q65-30");
    if (!col) return parse_token_line(mode, line, base_freq_hz, out);
    trim_ws(line);
    if (line.find("DecodeFinished") != std::string::npos) return CWSLG_SPOT_SKIP;
    std::string snr, dt, fq, msg;
    if (jt65) {                                                    // "HHMM snr  dt freq #  message" (OutputHandler.cpp:623-695)
        if (line.length() <= 27) return CWSLG_SPOT_SKIP;
        if (line[4] != ' ' || line[8] != ' ' || line[13] != ' ' || line[20] != ' ') return CWSLG_SPOT_SKIP;
        snr = line.substr(5, 3); dt = line.substr(9, 4); fq = line.substr(14, 4); msg = line.substr(22);
    } else {                                                       // FT8 / FT4 (:505-621), Q65 (:697-780): the same columns
        if (line.length() <= 28) return CWSLG_SPOT_SKIP;
        if (line[6] != ' ' || line[10] != ' ' || line[15] != ' ' || line[20] != ' ') return CWSLG_SPOT_SKIP;
        if (line[21] != '~' && line[21] != '+') return CWSLG_SPOT_SKIP;
        if (line[22] != ' ' || line[23] != ' ') return CWSLG_SPOT_SKIP;
        snr = line.substr(7, 3); dt = line.substr(11, 4); fq = line.substr(16, 4); msg = line.substr(24);
    }
    trim_ws(snr); trim_ws(dt); trim_ws(fq); trim_ws(msg);
    char *e1 = nullptr, *e2 = nullptr, *e3 = nullptr;
    const double f = std::strtod(fq.c_str(), &e1);                              // std::stod / stoi / stof throw on garbage:
    const long s = std::strtol(snr.c_str(), &e2, 10);                            // the reference then logs and drops the line
    const float d = std::strtof(dt.c_str(), &e3);
    if (e1 == fq.c_str() || e2 == snr.c_str() || e3 == dt.c_str()) return CWSLG_SPOT_SKIP;
    out->snr_db = (int32_t)s;
    out->dt_s = d;
    out->freq_hz = (uint32_t)(f + (double)base_freq_hz);
    std::string text = msg;
    if (std::strcmp(mode, "FT8") == 0) {                                          // Fox/Hound: the part after ';' names the sender
        const size_t semi = msg.find(';');
        if (semi != std::string::npos) text = msg.substr(semi + 1);
    }
    std::string call, loc; bool has_loc = false;
    const bool ok = message_to_spot(text, call, loc, has_loc);
    std::strncpy(out->message, msg.c_str(), sizeof(out->message) - 1);
    if (!ok) return CWSLG_SPOT_UNHANDLED;
    std::strncpy(out->call, call.c_str(), sizeof(out->call) - 1);
    if (has_loc) std::strncpy(out->locator, loc.c_str(), sizeof(out->locator) - 1);
    out->has_locator = has_loc ? 1 : 0;
    return CWSLG_SPOT_OK;
}

}  // namespace host
}  // namespace cwslg

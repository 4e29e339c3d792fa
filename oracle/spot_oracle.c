/*
 * spot_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as cwsl_oracle.h).
 *
 * CPU restatement of the FT8/FT4 decoder-output text stage: source/OutputHandler.cpp:505-621
 * (parseOutputFT4FT8, one line at a time), :924-1128 (handleMessageUniversal), :788-874 (parseCall / isCallPacked /
 * checkCall), :889-922 (isSOTAMATMessage), source/HamUtils.hpp:26-43 (isValidLocator), source/StringUtils.hpp:11-28.
 * PARITY: OutputHandler.cpp needs <windows.h>/Boost -- it cannot be compiled here and has no tests: "parity unpinned"
 * beyond this literal restatement (the reporter back ends and the ignore list are left out on both sides).
 * Written with C strings and explicit indices, on purpose unlike the product's std::string code.
 */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int32_t snr_db; float dt_s; uint32_t freq_hz; int32_t has_locator;
    char call[16]; char locator[8]; char message[64];
    int32_t drift, dbm;
} orc_spot_t;

static void trim(char *s)
{
    size_t n = strlen(s), a = 0;
    while (a < n && isspace((unsigned char)s[a])) ++a;
    while (n > a && isspace((unsigned char)s[n - 1])) --n;
    memmove(s, s + a, n - a);
    s[n - a] = 0;
}
static int valid_locator(const char *l)                              /* HamUtils.hpp:26-43 */
{
    if (strlen(l) != 4) return 0;
    if (!isalpha((unsigned char)l[0])) return 0;
    if (!isalpha((unsigned char)l[1])) return 0;
    if (!isdigit((unsigned char)l[2])) return 0;
    if (!isdigit((unsigned char)l[3])) return 0;
    return 1;
}
static int is_packed(const char *c)                                  /* :797-799 */
{
    const size_t n = strlen(c);
    return n >= 5 && c[0] == '<' && c[n - 1] == '>';
}
static void parse_call(char *c)                                      /* :788-795 */
{
    if (is_packed(c)) { const size_t n = strlen(c); memmove(c, c + 1, n - 2); c[n - 2] = 0; }
}
static int check_call(const char *c)                                 /* :802-874 (no ignore list) */
{
    const size_t n = strlen(c);
    if (n < 3) return 0;
    size_t letters = 0;
    for (size_t k = 0; k < n; ++k) if (isalpha((unsigned char)c[k])) letters++;
    if (letters == n) return 0;
    else if (letters == 0) return 0;
    if (strchr(c, ' ')) return 0;
    if (strchr(c, '.')) return 0;
    if (strchr(c, '+')) return 0;
    if (strchr(c, '-')) return 0;
    if (strchr(c, '?')) return 0;
    if (strchr(c, ';')) return 0;
    if (strchr(c, '=')) return 0;
    if (strchr(c, '~')) return 0;
    if (n == 4 && isalpha((unsigned char)c[0]) && isalpha((unsigned char)c[1]) && isdigit((unsigned char)c[2]) && isdigit((unsigned char)c[3])) return 0;
    return 1;
}
static int is_sotamat(const char *prefix, const char *call_sfx)      /* :889-922 */
{
    if (strlen(prefix) + strlen(call_sfx) + 1 != 13) return 0;
    const char *p[] = {"S", "SM", "STM", "STMT", "SOTAM", "SOTAMT", "SOTAMAT"};
    int found = 0;
    for (int k = 0; k < 7; ++k) if (!strcmp(p[k], prefix)) found = 1;
    if (!found) return 0;
    const char *slash = strchr(call_sfx, '/');
    if (!slash) return 0;
    const size_t sl = strlen(slash + 1);
    if (sl < 2) return 0;
    if (sl > 4) return 0;
    char base[64];
    const size_t bl = (size_t)(slash - call_sfx);
    memcpy(base, call_sfx, bl); base[bl] = 0;
    return check_call(base);
}
static void sub(char *dst, const char *s, size_t pos, size_t len)     /* std::string::substr with clamping */
{
    const size_t n = strlen(s);
    if (pos > n) pos = n;
    if (len > n - pos) len = n - pos;
    memcpy(dst, s + pos, len); dst[len] = 0;
}

/* handleMessageUniversal: 1 = reporter called with (call[, loc]); 0 = "Message not handled" */
static int handle_message(const char *in, char *o_call, char *o_loc, int *has_loc)
{
    char msg[256], a[256], b[256];
    snprintf(msg, sizeof msg, "%s", in);
    o_call[0] = 0; o_loc[0] = 0; *has_loc = 0;
    trim(msg);
    const char *chop[] = {"?", "a1", "a2", "q0", "q1", "q2", "q3", "q4", "q5"};
    for (int k = 0; k < 9; ++k) {
        char *q = strstr(msg, chop[k]);
        if (q) { *q = 0; trim(msg); }
    }
    const size_t len = strlen(msg);
    if (len < 6) return 0;
    size_t sp[64]; size_t ns = 0;
    for (size_t k = 0; k < len; ++k) if (msg[k] == ' ' && ns < 64) sp[ns++] = k;
    if (ns == 0) return 0;
    const int isCQ = msg[0] == 'C' && msg[1] == 'Q';
    if (isCQ && ns == 1 && msg[2] == ' ') {
        sub(a, msg, 3, len - 3); parse_call(a);
        if (check_call(a)) { strcpy(o_call, a); return 1; }
    } else if (isCQ && ns == 2) {
        sub(a, msg, sp[0] + 1, sp[1] - sp[0] - 1); parse_call(a);
        sub(b, msg, sp[1] + 1, len - sp[1] + 1);
        if (check_call(a)) {
            strcpy(o_call, a);
            if (valid_locator(b)) { strcpy(o_loc, b); *has_loc = 1; }
            return 1;
        } else {
            parse_call(b);
            if (check_call(b)) { strcpy(o_call, b); return 1; }
        }
    } else if (isCQ && ns == 3) {
        sub(a, msg, sp[1] + 1, (sp[2] - sp[1]) - 1); parse_call(a);
        sub(b, msg, sp[2] + 1, len - sp[2] + 1);
        if (check_call(a) && valid_locator(b)) { strcpy(o_call, a); strcpy(o_loc, b); *has_loc = 1; return 1; }
    } else if (!isCQ) {
        if (ns == 1) {
            sub(a, msg, sp[0] + 1, len); parse_call(a);
            sub(b, msg, 0, sp[0]);
            if (is_packed(b) && check_call(a)) { strcpy(o_call, a); return 1; }
            else if (is_sotamat(b, a)) { strcpy(o_call, a); return 1; }
        } else if (ns == 2) {
            sub(a, msg, sp[0] + 1, sp[1] - sp[0] - 1); parse_call(a);
            if (check_call(a)) { strcpy(o_call, a); return 1; }
        } else if (ns == 3) {
            sub(a, msg, sp[0] + 1, sp[1] - sp[0] - 1); parse_call(a);
            if (sp[2] - sp[1] == 2 && msg[sp[2] - 1] == 'R') {
                sub(b, msg, sp[2] + 1, len - sp[2] + 1);
                if (check_call(a) && valid_locator(b)) { strcpy(o_call, a); strcpy(o_loc, b); *has_loc = 1; return 1; }
            } else if (sp[2] - sp[1] == 4) {
                if (check_call(a)) { strcpy(o_call, a); return 1; }
            }
        }
    }
    return 0;
}

/* splitStringByDelim(line, ' ', true): StringUtils.hpp:48-68 */
static int split(const char *line, char tok[][64], int max)
{
    int n = 0;
    const char *p = line;
    while (*p) {
        while (*p == ' ') ++p;
        if (!*p) break;
        const char *e = p;
        while (*e && *e != ' ') ++e;
        if (n < max) { size_t l = (size_t)(e - p); if (l > 63) l = 63; memcpy(tok[n], p, l); tok[n][l] = 0; }
        ++n;
        p = e;
    }
    return n;
}
static int isnum(const char *s) { char *e; strtod(s, &e); return e != s; }

/* one line of parseOutputWSPR (:314-402), parseOutputFST4W (:152-240) or parseOutputFST4 (:243-312) */
static int token_line(const char *mode, const char *line_in, int64_t base_freq, orc_spot_t *out)
{
    char line[512], tok[16][64];
    snprintf(line, sizeof line, "%s", line_in);
    trim(line);
    const int n = split(line, tok, 16);
    const int wspr = !strcmp(mode, "WSPR"), fst4w = !strncmp(mode, "FST4W-", 6);
    if (wspr) {
        if (n != 8) return 2;
    } else {
        if (n < (fst4w ? 8 : 4) || strlen(line) <= 22) return 2;          /* lineVec[k] / line.at(k) would be out of range */
        if (line[18] != ' ') return 2;
        if (line[19] != '`') return 2;
        if (line[20] != ' ') return 2;
        if (line[21] != ' ') return 2;
    }
    if (!isnum(tok[1]) || !isnum(tok[2]) || !isnum(tok[3])) return 2;      /* std::stoi / stof / stod would throw */
    out->snr_db = (int32_t)strtol(tok[1], NULL, 10);
    out->dt_s = strtof(tok[2], NULL);
    double freqHz = (double)base_freq;
    freqHz += wspr ? strtod(tok[3], NULL) * 1000000.0 : strtod(tok[3], NULL);
    out->freq_hz = (uint32_t)freqHz;
    if (wspr || fst4w) {
        char call[64];
        snprintf(call, sizeof call, "%s", tok[5]);
        if (wspr) {
            parse_call(call);
            if (!isnum(tok[4]) || !isnum(tok[7])) return 2;
            out->drift = (int32_t)strtol(tok[4], NULL, 10);
        } else if (!isnum(tok[7])) return 2;
        out->dbm = (int32_t)strtol(tok[7], NULL, 10);
        if (!check_call(call)) return 1;
        snprintf(out->call, sizeof out->call, "%.15s", call);
        snprintf(out->locator, sizeof out->locator, "%.7s", tok[6]);
        out->has_locator = 1;
        return 0;
    }
    char msg[256], c[256], l[256]; int has = 0;
    sub(msg, line, 22, strlen(line) - 22); trim(msg);
    snprintf(out->message, sizeof out->message, "%.63s", msg);
    if (!handle_message(msg, c, l, &has)) return 1;
    snprintf(out->call, sizeof out->call, "%.15s", c);
    if (has) snprintf(out->locator, sizeof out->locator, "%.7s", l);
    out->has_locator = has;
    return 0;
}

/* one line of parseOutputFT4FT8: 0 ok, 1 unhandled, 2 skipped */
int orc_parse_decode_line(const char *mode, const char *line_in, int64_t base_freq, orc_spot_t *out)
{
    char line[512], f[16], msg[256];
    memset(out, 0, sizeof *out);
    const int jt65 = !strcmp(mode, "JT65");
    if (!jt65 && strcmp(mode, "FT8") && strcmp(mode, "FT4") && strcmp(mode, "Q65-30")) return token_line(mode, line_in, base_freq, out);
    snprintf(line, sizeof line, "%s", line_in);
    trim(line);
    if (strstr(line, "DecodeFinished")) return 2;
    if (jt65) {                                           /* parseOutputJT65, :623-695 */
        if (strlen(line) <= 27) return 2;
        if (line[4] != ' ') return 2;
        sub(f, line, 5, 3); trim(f);
        char *e; const long snr = strtol(f, &e, 10); if (e == f) return 2;
        if (line[8] != ' ') return 2;
        sub(f, line, 9, 4); trim(f);
        const float dt = strtof(f, &e); if (e == f) return 2;
        if (line[13] != ' ') return 2;
        sub(f, line, 14, 4); trim(f);
        const double fq = strtod(f, &e); if (e == f) return 2;
        if (line[20] != ' ') return 2;
        if (line[19] != '#' && line[20] != ' ') return 2;
        sub(msg, line, 22, strlen(line)); trim(msg);
        out->snr_db = (int32_t)snr; out->dt_s = dt; out->freq_hz = (uint32_t)(fq + (double)base_freq);
        snprintf(out->message, sizeof out->message, "%.63s", msg);
        char call[256], loc[256]; int has = 0;
        if (!handle_message(msg, call, loc, &has)) return 1;
        snprintf(out->call, sizeof out->call, "%.15s", call);
        if (has) snprintf(out->locator, sizeof out->locator, "%.7s", loc);
        out->has_locator = has;
        return 0;
    }
    if (strlen(line) <= 28) return 2;
    if (line[6] != ' ') return 2;
    sub(f, line, 7, 3); trim(f);
    char *e; const long snr = strtol(f, &e, 10); if (e == f) return 2;
    if (line[10] != ' ') return 2;
    sub(f, line, 11, 4); trim(f);
    const float dt = strtof(f, &e); if (e == f) return 2;
    if (line[15] != ' ') return 2;
    sub(f, line, 16, 4); trim(f);
    const double fq = strtod(f, &e); if (e == f) return 2;
    if (line[20] != ' ') return 2;
    if (line[21] != '~' && line[21] != '+') return 2;
    if (line[22] != ' ') return 2;
    if (line[23] != ' ') return 2;
    sub(msg, line, 24, strlen(line) - 24); trim(msg);
    const double actual = fq + (double)base_freq;
    out->snr_db = (int32_t)snr; out->dt_s = dt; out->freq_hz = (uint32_t)actual;
    snprintf(out->message, sizeof out->message, "%.63s", msg);
    const char *text = msg;
    const char *semi = strchr(msg, ';');
    if (!strcmp(mode, "FT8") && semi) text = semi + 1;             /* Fox/Hound: only the second part names the sender */
    char call[256], loc[256]; int has = 0;
    if (!handle_message(text, call, loc, &has)) return 1;
    snprintf(out->call, sizeof out->call, "%.15s", call);
    if (has) snprintf(out->locator, sizeof out->locator, "%.7s", loc);
    out->has_locator = has;
    return 0;
}

/*
 * handoff_oracle.c -- TEST INFRASTRUCTURE ONLY (same rules as cwsl_oracle.h).
 *
 * CPU restatement of the decoder hand-off formats downstream of the hot path:
 *   - the shared-memory block jt9 maps (source/DecoderPool.hpp:58-108, filled at :451-590),
 *   - the js8 variant (:110-171, filled at :760-804),
 *   - the decoder command lines (:634-659 shared memory, :1007-1046 wave file) and the route choice (:379-395).
 *
 * PARITY: DecoderPool.hpp includes <windows.h>/Qt and cannot be compiled here, and the reference has no tests
 * for it, so this restatement is pinned only by the interface itself (the struct is an ABI shared with
 * WSJT-X's lib/jt9com.f90) -- "parity unpinned" beyond that.  Here the layouts are plain C structs laid out by
 * the compiler; the product derives the same offsets from a field table, and tests compare the bytes.
 */
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ORC_NSMAX 6827
#define ORC_D2    (30 * 60 * 12000)

struct orc_jt9_params {
    int nutc; bool ndiskdat; int ntrperiod; int nQSOProgress; int nfqso; int nftx; bool newdat; int npts8;
    int nfa; int nfSplit; int nfb; int ntol; int kin; int nzhsym; int nsubmode; bool nagain; int ndepth;
    bool lft8apon; bool lapcqonly; bool ljt65apon; int napwid; int ntxmode; int nmode; int minw; bool nclearave;
    int minSync; float emedelay; float dttol; int nlist; int listutc[10]; int n2pass; int nranera; int naggressive;
    bool nrobust; int nexp_decode; char datetime[20]; char mycall[12]; char mygrid[6]; char hiscall[12]; char hisgrid[6];
};
typedef struct {
    int ipc[3];
    float ss[184 * ORC_NSMAX];
    float savg[ORC_NSMAX];
    float sred[5760];
    short d2[ORC_D2];
    struct orc_jt9_params params;
} orc_jt9_block;

struct orc_js8_params {
    int nutc; bool ndiskdat; int ntrperiod; int nQSOProgress; int nfqso; int nftx; bool newdat; int npts8;
    int nfa; int nfb; int ntol; bool syncStats; int kin; int kposA, kposB, kposC, kposE, kposI;
    int kszA, kszB, kszC, kszE, kszI; int nzhsym; int nsubmode; int nsubmodes; bool nagain; int ndepth;
    bool lft8apon; bool lapcqonly; bool ljt65apon; int napwid; int ntxmode; int nmode; int minw; bool nclearave;
    int minSync; float emedelay; float dttol; int nlist; int listutc[10]; int n2pass; int nranera; int naggressive;
    bool nrobust; int nexp_decode; char datetime[20]; char mycall[12]; char mygrid[6]; char hiscall[12]; char hisgrid[6];
    int ndebug;
};
typedef struct {
    float ss[184 * ORC_NSMAX];
    float savg[ORC_NSMAX];
    float sred[5760];
    short d2[ORC_D2];
    struct orc_js8_params params;
} orc_js8_block;

size_t orc_jt9_block_bytes(void) { return sizeof(orc_jt9_block); }
size_t orc_js8_block_bytes(void) { return sizeof(orc_js8_block); }

#define OFF9(m) if (!strcmp(name, #m)) return (long)offsetof(orc_jt9_block, params.m)
#define OFF8(m) if (!strcmp(name, #m)) return (long)offsetof(orc_js8_block, params.m)
long orc_jt9_offset(const char *name)
{
    if (!strcmp(name, "ipc")) return (long)offsetof(orc_jt9_block, ipc);
    if (!strcmp(name, "ss")) return (long)offsetof(orc_jt9_block, ss);
    if (!strcmp(name, "savg")) return (long)offsetof(orc_jt9_block, savg);
    if (!strcmp(name, "sred")) return (long)offsetof(orc_jt9_block, sred);
    if (!strcmp(name, "d2")) return (long)offsetof(orc_jt9_block, d2);
    if (!strcmp(name, "params")) return (long)offsetof(orc_jt9_block, params);
    OFF9(nutc); OFF9(ndiskdat); OFF9(ntrperiod); OFF9(nQSOProgress); OFF9(nfqso); OFF9(nftx); OFF9(newdat); OFF9(npts8);
    OFF9(nfa); OFF9(nfSplit); OFF9(nfb); OFF9(ntol); OFF9(kin); OFF9(nzhsym); OFF9(nsubmode); OFF9(nagain); OFF9(ndepth);
    OFF9(lft8apon); OFF9(lapcqonly); OFF9(ljt65apon); OFF9(napwid); OFF9(ntxmode); OFF9(nmode); OFF9(minw); OFF9(nclearave);
    OFF9(minSync); OFF9(emedelay); OFF9(dttol); OFF9(nlist); OFF9(listutc); OFF9(n2pass); OFF9(nranera); OFF9(naggressive);
    OFF9(nrobust); OFF9(nexp_decode); OFF9(datetime); OFF9(mycall); OFF9(mygrid); OFF9(hiscall); OFF9(hisgrid);
    return -1;
}
long orc_js8_offset(const char *name)
{
    if (!strcmp(name, "ss")) return (long)offsetof(orc_js8_block, ss);
    if (!strcmp(name, "savg")) return (long)offsetof(orc_js8_block, savg);
    if (!strcmp(name, "sred")) return (long)offsetof(orc_js8_block, sred);
    if (!strcmp(name, "d2")) return (long)offsetof(orc_js8_block, d2);
    if (!strcmp(name, "params")) return (long)offsetof(orc_js8_block, params);
    OFF8(nutc); OFF8(ndiskdat); OFF8(ntrperiod); OFF8(nQSOProgress); OFF8(nfqso); OFF8(nftx); OFF8(newdat); OFF8(npts8);
    OFF8(nfa); OFF8(nfb); OFF8(ntol); OFF8(syncStats); OFF8(kin); OFF8(kposA); OFF8(kposB); OFF8(kposC); OFF8(kposE);
    OFF8(kposI); OFF8(kszA); OFF8(kszB); OFF8(kszC); OFF8(kszE); OFF8(kszI); OFF8(nzhsym); OFF8(nsubmode); OFF8(nsubmodes);
    OFF8(nagain); OFF8(ndepth); OFF8(lft8apon); OFF8(lapcqonly); OFF8(ljt65apon); OFF8(napwid); OFF8(ntxmode); OFF8(nmode);
    OFF8(minw); OFF8(nclearave); OFF8(minSync); OFF8(emedelay); OFF8(dttol); OFF8(nlist); OFF8(listutc); OFF8(n2pass);
    OFF8(nranera); OFF8(naggressive); OFF8(nrobust); OFF8(nexp_decode); OFF8(datetime); OFF8(mycall); OFF8(mygrid);
    OFF8(hiscall); OFF8(hisgrid); OFF8(ndebug);
    return -1;
}

static int starts(const char *s, const char *pre) { return strncmp(s, pre, strlen(pre)) == 0; }

/* DecoderPool.hpp:451-590.  Returns 0, or -5 for "Unknown mode". */
int orc_jt9_fill(void *block, const char *mode, int decodedepth, int highest_hz, const int16_t *audio, size_t nel)
{
    orc_jt9_block *d = (orc_jt9_block *)block;
    memset(d, 0, sizeof *d);
    d->params.nfa = 0;
    d->params.nfb = highest_hz;
    d->params.ndepth = decodedepth;
    d->params.nutc = 0;
    d->params.newdat = 1;
    d->params.nagain = 0;
    d->params.emedelay = 0;
    d->params.nrobust = 0;
    d->params.ndiskdat = 0;
    d->params.minw = 0;
    d->params.minSync = 0;
    d->params.dttol = 4;
    if (!strcmp(mode, "FT8")) {
        d->params.lft8apon = true; d->params.nzhsym = 0; d->params.nmode = 8; d->params.napwid = 50; d->params.ntrperiod = 15;
    } else if (!strcmp(mode, "FT4")) {
        d->params.nmode = 5; d->params.ntrperiod = (int)7.5; d->params.napwid = 80; d->params.nzhsym = 0;
    } else if (!strcmp(mode, "Q65-30")) {
        d->params.nmode = 66; d->params.ntxmode = 66; d->params.ntrperiod = (int)30.0; d->params.nzhsym = 196;
    } else if (!strcmp(mode, "JT65")) {
        d->params.nzhsym = 174; d->params.ntxmode = 65; d->params.nmode = 65; d->params.ntrperiod = 60;
    } else if (starts(mode, "FST4-")) {
        int per = 0, nz = 0, nfa = 900;
        if (!strcmp(mode, "FST4-60")) { per = 60; nz = 187; }
        else if (!strcmp(mode, "FST4-120")) { per = 120; nz = 387; }
        else if (!strcmp(mode, "FST4-300")) { per = 300; nz = 1003; nfa = 700; }
        else if (!strcmp(mode, "FST4-900")) { per = 900; nz = 3107; }
        else if (!strcmp(mode, "FST4-1800")) { per = 1800; nz = 6232; }
        else return -5;
        d->params.ndepth = 1; d->params.nfa = nfa; d->params.nfb = 1100; d->params.nzhsym = nz; d->params.nmode = 240;
        d->params.ntol = 100; d->params.ntrperiod = per;
    } else if (starts(mode, "FST4W-")) {
        int per = 0, nz = 0;
        if (!strcmp(mode, "FST4W-120")) { per = 120; nz = 387; }
        else if (!strcmp(mode, "FST4W-300")) { per = 300; nz = 1003; }
        else if (!strcmp(mode, "FST4W-900")) { per = 900; nz = 3107; }
        else if (!strcmp(mode, "FST4W-1800")) { per = 1800; nz = 6232; }
        else return -5;
        d->params.nzhsym = nz; d->params.nmode = 241; d->params.ntol = 100; d->params.ntrperiod = per;
        d->params.nfqso = 1500; d->params.nexp_decode = 256 * 3;
    } else {
        return -5;
    }
    d->ipc[0] = d->params.nzhsym;
    d->ipc[1] = 1;
    d->ipc[2] = -1;
    if (nel > (size_t)ORC_D2) nel = ORC_D2;
    memcpy(d->d2, audio, nel * sizeof(int16_t));
    return 0;
}

/* DecoderPool.hpp:760-804 */
int orc_js8_fill(void *block, int decodedepth, int highest_hz, const int16_t *audio, size_t nel)
{
    orc_js8_block *d = (orc_js8_block *)block;
    memset(d, 0, sizeof *d);
    d->params.nfa = 0;
    d->params.nfb = highest_hz;
    d->params.ndepth = decodedepth;
    d->params.newdat = 1;
    d->params.dttol = 4;
    d->params.syncStats = false;
    d->params.ntrperiod = -1;
    d->params.nsubmode = -1;
    d->params.n2pass = 1;
    d->params.npts8 = 50 * 6912 / 16;
    d->params.kszA = ORC_D2 - 1;
    d->params.kposA = 0;
    d->params.nsubmodes = 1;
    d->params.lft8apon = false;
    d->params.nzhsym = 0;
    d->params.nmode = 8;
    d->params.napwid = 50;
    if (nel > (size_t)ORC_D2) nel = ORC_D2;
    memcpy(d->d2, audio, nel * sizeof(int16_t));
    return 0;
}

/* DecoderPool.hpp:379-395 : 1 shared memory, 0 wave file */
int orc_decoder_route(const char *mode, int transfer_shmem)
{
    if (!strcmp(mode, "WSPR")) return 0;
    if (!transfer_shmem) return 0;
    if (!strcmp(mode, "JS8")) return 0;
    if (starts(mode, "FST4-") || starts(mode, "FST4W-")) return 0;
    return 1;
}

/* DecoderPool.hpp:634-659 (shmem) and :1007-1046 (file).  Writes app and opts; returns 0 or -5. */
int orc_decoder_command(const char *mode, int shmem_route, int threads, int depth, int highest_hz, int wspr_cycles,
                        float trperiod, const char *target, char *app, size_t app_cap, char *opts, size_t opts_cap)
{
    char op[512];
    const char *prog = "jt9.exe";
    if (shmem_route) {
        if (!strcmp(mode, "FT8")) snprintf(op, sizeof op, " -8 -m %d ", threads);
        else if (!strcmp(mode, "FT4")) snprintf(op, sizeof op, " -5 -m %d ", threads);
        else if (!strcmp(mode, "Q65-30")) snprintf(op, sizeof op, " -3 -m %d -p 30 -H %d ", threads, highest_hz);
        else if (!strcmp(mode, "JT65")) snprintf(op, sizeof op, " -6 -m %d ", threads);
        else if (starts(mode, "FST4W-")) snprintf(op, sizeof op, " -W -m %d ", threads);
        else if (starts(mode, "FST4-")) snprintf(op, sizeof op, " -7 -m %d ", threads);
        else return -5;
        snprintf(opts, opts_cap, "%s -s %s", op, target);
    } else {
        if (!strcmp(mode, "FT8")) snprintf(op, sizeof op, " -8 -m %d -d %d -w 1 -H %d ", threads, depth, highest_hz);
        else if (!strcmp(mode, "FT4")) snprintf(op, sizeof op, " -5 -m %d -d %d -w 1 -H %d ", threads, depth, highest_hz);
        else if (!strcmp(mode, "Q65-30")) snprintf(op, sizeof op, " -3 -p 30 -H %d ", highest_hz);
        else if (!strcmp(mode, "WSPR")) { prog = "wsprd.exe"; snprintf(op, sizeof op, " -C %d -o 5 -d ", wspr_cycles); }
        else if (!strcmp(mode, "JT65")) snprintf(op, sizeof op, " -6 -d %d ", depth);
        else if (starts(mode, "FST4W-")) snprintf(op, sizeof op, " -W -p %d -m %d -d %d -L 1400 -H 1600 -F 200 ", (int)trperiod, threads, depth);
        else if (starts(mode, "FST4-")) snprintf(op, sizeof op, " -7 -p %d -m %d ", (int)trperiod, threads);
        else if (!strcmp(mode, "JS8")) { prog = "js8.exe"; snprintf(op, sizeof op, " -8 -m %d ", threads); }
        else return -5;
        snprintf(opts, opts_cap, "%s%s", op, target);
    }
    snprintf(app, app_cap, "%s", prog);
    return 0;
}

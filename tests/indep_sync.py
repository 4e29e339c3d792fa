"""An INDEPENDENT second implementation of the candidate SELECTION logic of rows a13 / a14, CPU only, numpy on float64 FFTs -- written from the
published algorithms as SURVEY.md section 8c summarises them (WSJT-X 2.x: lib/ft8/sync8.f90, lib/ft4/getcandidates4.f90 + ft4_baseline.f90,
wsprd.c's candidate pick, lib/fst4_decode.f90's get_candidates_fst4), NOT from oracle/*.c: whole-array numpy where the restatement loops, double
precision where it is single, numpy's FFT and polyfit where it has its own.  tests/test_indep_sync.py compares the (bin, lag) keys the two produce.

It is not a pin (the upstream source is not in this container: parity of a13 / a14 stays "unpinned"); it is evidence that the restatement and the
kernels -- which agree with each other bit for bit -- do not share one author's misreading of the selection rules: lag windows, first-maximum
maxloc, the 40th-percentile normalisation, the pre-candidate limit, the near-duplicate rule, the smoothing / baseline / peak pick of FT4, wsprd's
smoothed-spectrum peaks, and the CLEAN loop of FST4W."""
import numpy as np

ICOS7 = np.array([3, 1, 4, 0, 6, 5, 2])


def _nint(x):
    return int(np.floor(x + 0.5)) if x >= 0 else -int(np.floor(-x + 0.5))


# ------------------------------------------------------------------------------------------------------------------------------ FT8 (sync8)
def ft8_sync2d(frame_i16):
    """s(i, j) = |X_i|^2 of 372 quarter-symbol steps (1920 samples / 300, zero-padded to 3840) and the Costas metric
    sync2d(i, lag) = max(t / ((t0 - t) / 6)) over {all three arrays, the last two}, lag = -62..62, for every bin i."""
    nsps, nstep, nfft, nhsym, jz = 1920, 480, 3840, 372, 62
    dd = np.asarray(frame_i16[:180000], np.float64) / 300.0
    idx = np.arange(nhsym)[:, None] * nstep + np.arange(nsps)[None, :]
    s = np.abs(np.fft.rfft(dd[idx], nfft, axis=1)) ** 2                      # s[j, i], i = 0..1920 (i = 0 unused)
    nb = s.shape[1]
    # zero-padded time axis: index m (1-based step) -> column m - 1 + pad
    pad = 400
    sp = np.zeros((nb + 16, nhsym + 2 * pad))
    sp[:nb, pad:pad + nhsym] = s.T
    # a(i, m) = sum_n s(i + 2 icos7[n], m + 4 n);  b(i, m) = sum_n sum_k s(i + 2 k, m + 4 n)   (terms outside 1..372 are absent)
    tones = np.zeros((nb, sp.shape[1]))                                      # sum over the seven tone bins of one step
    for k in range(7):
        tones += sp[2 * k:2 * k + nb]
    a = np.zeros((nb, sp.shape[1] - 24))
    b = np.zeros_like(a)
    for n in range(7):
        a += sp[2 * ICOS7[n]:2 * ICOS7[n] + nb, 4 * n:4 * n + a.shape[1]]
        b += tones[:, 4 * n:4 * n + a.shape[1]]
    jstrt = 12                                                               # int(0.5 / 0.04)
    lags = np.arange(-jz, jz + 1)
    col = lags + jstrt - 1 + pad                                             # m = lag + jstrt (1-based) -> column
    ta, tb, tc = a[:, col], a[:, col + 144], a[:, col + 288]
    t0a, t0b, t0c = b[:, col], b[:, col + 144], b[:, col + 288]
    with np.errstate(divide="ignore", invalid="ignore"):
        t, t0 = ta + tb + tc, t0a + t0b + t0c
        abc = t / ((t0 - t) / 6.0)
        t, t0 = tb + tc, t0b + t0c
        bc = t / ((t0 - t) / 6.0)
    return np.fmax(abc, bc)                                                  # [bin, lag + 62]


def ft8_candidates(frame_i16, nfa=200, nfb=3000, syncmin=1.5, maxcand=200, max_pre=1000):
    """-> (list of (bin, lag, sync) in frequency order, dict of the per-bin arrays)."""
    df, tstep, jz = 3.125, 0.04, 62
    sync2d = ft8_sync2d(frame_i16)
    ia, ib = max(1, _nint(nfa / df)), _nint(nfb / df)
    bins = np.arange(ia, ib + 1)
    near = sync2d[bins][:, jz - 10:jz + 11]
    jpeak = np.argmax(near, axis=1) - 10                                     # first maximum
    red = near[np.arange(len(bins)), jpeak + 10]
    jpeak2 = np.argmax(sync2d[bins], axis=1) - jz
    red2 = sync2d[bins][np.arange(len(bins)), jpeak2 + jz]
    iz = len(bins)
    npct = _nint(0.40 * iz)
    red = red / np.sort(red)[npct - 1]
    red2 = red2 / np.sort(red2)[npct - 1]
    pre = []
    for r in np.argsort(-red, kind="stable")[:min(max_pre, iz)]:            # strongest first
        if len(pre) >= max_pre:
            break
        if red[r] >= syncmin:
            pre.append([bins[r], jpeak[r], red[r]])
        if jpeak2[r] == jpeak[r]:
            continue
        if len(pre) >= max_pre:
            break
        if red2[r] >= syncmin:
            pre.append([bins[r], jpeak2[r], red2[r]])
    f = np.array([p[0] * df for p in pre])
    # upstream's reals are single precision, and this comparison is where it shows: dt = (lag - 0.5) * tstep in float32, and whether two entries
    # ONE step apart are "closer than 0.04 s" depends on how those two products round (0.04 is not a binary fraction)
    dt = (np.array([p[1] for p in pre], np.float32) - np.float32(0.5)) * np.float32(480.0 / 12000.0)
    sy = np.array([p[2] for p in pre], np.float64)
    for i in range(1, len(pre)):                                             # near-duplicates: the weaker of two within 4 Hz and one step goes
        hits = np.nonzero((np.abs(np.abs(f[i]) - np.abs(f[:i])) < 4.0) & (np.abs(dt[i] - dt[:i]).astype(np.float32) < np.float32(0.04)))[0]
        for j in hits:
            if sy[i] >= sy[j]:
                sy[j] = 0.0
            if sy[i] < sy[j]:
                sy[i] = 0.0
    order = np.argsort(f, kind="stable")
    out = [(int(pre[k][0]), int(pre[k][1]), float(sy[k])) for k in order if sy[k] >= syncmin][:maxcand]
    return out, dict(bins=bins, red=red, red2=red2, jpeak=jpeak, jpeak2=jpeak2)


# ------------------------------------------------------------------------------------------------------------------------ FT4 (getcandidates4)
def nuttall(n):
    k = np.arange(n)
    return 0.3635819 - 0.4891775 * np.cos(2 * np.pi * k / n) + 0.1365995 * np.cos(4 * np.pi * k / n) - 0.0106411 * np.cos(6 * np.pi * k / n)


def ft4_candidates(frame_i16, fa=200.0, fb=4000.0, syncmin=1.2, maxcand=200):
    """-> (list of (bin, f_peak_hz, height), savsm normalised, sbase)."""
    nsps, nfft, nstep, nmax, nh1 = 576, 2304, 576, 72576, 1152
    df = 12000.0 / nfft
    dd = np.asarray(frame_i16[:nmax], np.float64) / 300.0
    nsym = min(122, (nmax - nfft) // nstep + 1)                              # do j = 1, NHSYM, leaving early if a window would end past the buffer
    idx = np.arange(nsym)[:, None] * nstep + np.arange(nfft)[None, :]
    s = np.abs(np.fft.rfft(dd[idx] * nuttall(nfft)[None, :], axis=1)) ** 2
    savg = s.sum(axis=0) / 122.0                                             # NHSYM = 122 (the steps past the buffer contribute nothing)
    savsm = np.zeros(nh1 + 1)
    c = np.concatenate(([0.0], np.cumsum(savg)))
    i = np.arange(8, nh1 - 6)
    savsm[i] = (c[i + 8] - c[i - 7]) / 15.0                                  # mean of savg[i-7 .. i+7]
    nfa = max(int(fa / df), 8)
    nfb = min(int(fb / df), _nint(4910.0 / df))
    # baseline (ft4_baseline): lowest 10 % of each of ten segments of the dB spectrum, 4th-order polynomial through them, + 0.65 dB
    ia, ib = max(_nint(200.0 / df), nfa), min(nh1, nfb)
    sdb = 10.0 * np.log10(savg)
    nlen = (ib - ia + 1) // 10
    i0 = (ib - ia + 1) // 2
    xs, ys = [], []
    for n in range(10):
        ja = ia + n * nlen
        seg = sdb[ja:ja + nlen]
        j = _nint(nlen * 0.10)
        base = np.sort(seg)[min(max(j, 1), nlen) - 1]
        for k in np.nonzero(seg <= base)[0]:
            if len(xs) < 1000:
                xs.append(ja + k - i0); ys.append(seg[k])
    coef = np.polynomial.polynomial.polyfit(np.array(xs, np.float64), np.array(ys), 4)
    tt = np.arange(ia, ib + 1) - i0
    sbase = np.zeros(nh1 + 1)
    sbase[ia:ib + 1] = 10.0 ** ((np.polynomial.polynomial.polyval(tt.astype(np.float64), coef) + 0.65) / 10.0)
    if np.any(sbase[nfa:nfb + 1] <= 0):
        return [], savsm, sbase
    savsm[nfa:nfb + 1] /= sbase[nfa:nfb + 1]
    f_offset = -1.5 * 12000.0 / nsps
    out = []
    for i in range(nfa + 1, nfb):
        if savsm[i] >= savsm[i - 1] and savsm[i] >= savsm[i + 1] and savsm[i] >= syncmin:
            den = savsm[i - 1] - 2 * savsm[i] + savsm[i + 1]
            dl = 0.5 * (savsm[i - 1] - savsm[i + 1]) / den if den != 0.0 else 0.0
            fpeak = (i + dl) * df + f_offset
            if fpeak < 200.0 or fpeak > 4910.0:
                continue
            out.append((i, fpeak, savsm[i] - 0.25 * (savsm[i - 1] - savsm[i + 1]) * dl))
            if len(out) == maxcand:
                break
    return out, savsm, sbase


# ----------------------------------------------------------------------------------------------------------------- WSPR (wsprd's candidate pick)
def wspr_pick(ps, fmin=-110.0, fmax=110.0):
    """ps[512][nffts] = power spectra of the 375 Hz baseband (bin 256 = 1500 Hz).  -> (list of (freq_hz, snr_db) strongest first, smspec[411])."""
    df = 375.0 / 256.0 / 2.0
    psavg = np.asarray(ps, np.float64).sum(axis=1)
    c = np.concatenate(([0.0], np.cumsum(psavg)))
    i = np.arange(411)
    lo = 256 - 205 + i - 3
    smspec = c[lo + 7] - c[lo]                                               # seven-bin window around 256 - 205 + i
    noise = np.sort(smspec)[122]                                             # 30th percentile of 411
    min_snr = 10.0 ** (-8.0 / 10.0)
    smspec = smspec / noise - 1.0
    smspec = np.where(smspec < min_snr, 0.1 * min_snr, smspec)
    pk = [(float((j - 205) * df), float(10.0 * np.log10(smspec[j]) - 26.3)) for j in range(1, 410)
          if smspec[j] > smspec[j - 1] and smspec[j] > smspec[j + 1]][:200]
    pk = [p for p in pk if fmin <= p[0] <= fmax]
    pk.sort(key=lambda p: -p[1])
    return pk, smspec


# -------------------------------------------------------------------------------------------------------- FST4W (get_candidates_fst4's CLEAN loop)
def fst4w_pick(power, band_first_bin, nfa=1400, nfb=1600, minsync=1.2, nsps=8200, nfft1=1440000, fs=12000.0, hmod=1):
    """power = |c|^2 of the long transform's bins from band_first_bin on.  -> (list of (bin, snr) in the order found, s2 normalised)."""
    df1 = fs / nfft1
    df2 = (fs / nsps) / 2.0
    nd = int(df2 / df1)
    ndh = nd // 2
    ina, inb = _nint(max(100.0, float(nfa)) / df2), _nint(min(4800.0, float(nfb)) / df2)
    ia, ib = ina, inb
    nnw = _nint(48000.0 * nsps * 2.0 / fs)
    p = np.asarray(power, np.float64)
    c = np.concatenate(([0.0], np.cumsum(p)))
    s = np.zeros(nnw + 1)
    for i in range(ina, inb + 1):
        j0 = _nint(i * df2 / df1) - band_first_bin
        s[i] = c[j0 + ndh + 1] - c[j0 - ndh]
    ina, inb = max(ina, 1 + 3 * hmod), min(inb, nnw - 3 * hmod)
    s2 = np.zeros(nnw + 1)
    i = np.arange(ina, inb + 1)
    s2[i] = s[i - 3 * hmod] + s[i - hmod] + s[i + hmod] + s[i + 3 * hmod]
    seg = np.sort(s2[ina + 3 * hmod:inb - 3 * hmod + 1])
    j = min(max(_nint(len(seg) * 0.30), 1), len(seg))
    s2 = s2 / seg[j - 1]
    norm = s2.copy()
    ia, ib = max(ia, 3), min(ib, nnw - 2)
    xdb = {-3: 0.25, -2: 0.50, -1: 0.75, 0: 1.0, 1: 0.75, 2: 0.50, 3: 0.25}
    out = []
    while len(out) < 100:
        ip = ia + int(np.argmax(s2[ia:ib + 1]))
        pv = s2[ip]
        if pv < minsync:
            break
        for d, w in xdb.items():
            k = ip + 2 * hmod * d
            if ia <= k <= ib:
                s2[k] = max(0.0, s2[k] - 0.9 * pv * w)
        out.append((ip, float(pv)))
    return out, norm

#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref = unmodified SSBD.hpp/LowPass.hpp
compiled in place with g++ -std=c++17 -O2 -ffp-contract=off).  Run in the build container only:

    make -C oracle ref && python tests/gen_golden.py

The fixtures hold inputs as seeds (the portable generator of oracle/cwsl_oracle.c regenerates them) and
expected outputs as raw bit patterns / samples / checksums -- data only, no reference source.
Rows a8/a9 (prepareAudio, int16) have no compilable reference (Instance.cpp needs Win32); their
expectations are this repo's restatement applied to the reference's SSBD output.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

CASES = {
    192000: [0, 1234, 24000, 87000, -50000, -93000, -26000],
    96000: [0, 1234, 24000, -45000],
    48000: [0, 1234, -20000, 17000],
}
SAMPLE_STRIDE = 997


def tones_for(f):
    return [f + 700.0, f + 1500.5, f + 2600.25]


def phasor_points(nblk):
    pts = [0, 1, 2, 15, 16, 17, 255, 256, 257]
    pts += list(range(4096, nblk, 4096))
    pts.append(nblk - 1)
    return np.array(sorted(set(p for p in pts if p < nblk)), np.int64)


def gen_constants():
    for fs, freqs in CASES.items():
        for f in freqs:
            r = O.RefDemod(fs, float(np.float32(f)))
            tone = r.tone
            np.savez_compressed(
                os.path.join(OUT, f"const_fs{fs}_f{f}.npz"),
                fs=fs, f=f, block=r.block, ntaps=r.ntaps,
                taps_bits=r.taps.view(np.uint32), tone_bits=tone.view(np.uint32),
                inc_bits=np.array([r.phase_inc]).view(np.uint32))


def gen_slot(name, mode, fs, f, seed, n_iq, iq_len, scale_ft=0.90, scale_wspr=0.20):
    D = fs // 12000
    iq = O.synth_iq(seed, n_iq, fs, tones_hz=tones_for(f), amp=2.0e4)
    r = O.RefDemod(fs, float(np.float32(f)))
    audio, trace = r.run(iq, trace=True)
    nblk = n_iq // D
    pts = phasor_points(nblk)
    frame = np.zeros(O.frame_len(mode), np.float32)
    frame[:nblk] = audio
    scaled, factor, peak = O.prepare_audio(frame, mode, scale_ft, scale_wspr)
    i16 = O.to_int16(scaled)
    np.savez_compressed(
        os.path.join(OUT, f"slot_{name}.npz"),
        mode=mode, fs=fs, f=f, seed=seed, n_iq=n_iq, iq_len=iq_len, tones=np.array(tones_for(f)), amp=2.0e4,
        phasor_idx=pts, phasor_bits=trace[pts].view(np.uint64),
        audio_head_bits=audio[:4096].view(np.uint32), audio_tail_bits=audio[-512:].view(np.uint32),
        audio_every_bits=audio[::SAMPLE_STRIDE].view(np.uint32),
        audio_maxabs_bits=np.array([np.abs(audio).max()], np.float32).view(np.uint32),
        audio_checksum=O.checksum(audio),
        peak_bits=np.array([peak], np.float32).view(np.uint32),
        factor_bits=np.array([factor], np.float32).view(np.uint32),
        i16_crc32=O.crc32(i16), i16_head=i16[:256], i16_tail=i16[nblk - 256:nblk], i16_len=len(i16), n_valid=nblk)
    print(name, "peak", float(peak), "factor", float(factor), "crc %08x" % O.crc32(i16))


def gen_adversarial():
    """tests/adversarial.py's inputs through the reference's own SSBD (oracle/_ref): the whole float frame as CRC32 + samples, the
    restated prepareAudio / int16 on top (rows a8/a9 have no compilable reference)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import adversarial as A
    for name in sorted(A.CASES):
        iq = A.make_iq(name)
        r = O.RefDemod(A.FS, float(np.float32(A.F)))
        audio = r.run(iq)
        nblk = len(iq) // 16
        frame = np.zeros(O.frame_len("FT8"), np.float32)
        frame[:nblk] = audio
        scaled, factor, peak = O.prepare_audio(frame, "FT8", 0.90, 0.20)
        i16 = O.to_int16(scaled)
        np.savez_compressed(
            os.path.join(OUT, f"adv_{name}.npz"), name=name, iq_crc32=A.iq_crc(iq), n_valid=nblk,
            audio_crc32=O.crc32(audio.view(np.uint32)), audio_head_bits=audio[:2048].view(np.uint32),
            audio_every_bits=audio[::97].view(np.uint32),
            peak_bits=np.array([peak], np.float32).view(np.uint32), factor_bits=np.array([factor], np.float32).view(np.uint32),
            i16_crc32=O.crc32(i16), i16_head=i16[:256], argmax=int(np.abs(audio).argmax()),
            input_peak=float(np.abs(np.concatenate([iq.real, iq.imag])).max()))
        print("adv", name, "frame peak %.6g at output %d, input peak %.6g, factor %.6g, i16 crc %08x"
              % (float(peak), int(np.abs(audio).argmax()), float(np.abs(iq).max()), float(factor), O.crc32(i16)))


def main():
    if not O.have_ref():
        raise SystemExit("oracle/_ref/libcwsl_ref.so missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(OUT, exist_ok=True)
    gen_constants()
    # (4)/(5): one full FT8 slot per 192 kHz channel (BASELINE config 1 is f = -26000)
    for f in CASES[192000]:
        gen_slot(f"ft8_fs192000_f{f}", "FT8", 192000, f, 0xC0FFEE ^ (f & 0xFFFF), 2880000 // 2048 * 2048, 2048)
    # other rates, shorter
    gen_slot("ft8_fs96000_f1234", "FT8", 96000, 1234, 0xBEEF, 1440000 // 1024 * 1024, 1024)
    gen_slot("ft8_fs48000_f-20000", "FT8", 48000, -20000, 0xF00D, 720000 // 512 * 512, 512)
    # (6): one FT4 and one 120 s frame (WSPR uses the 0.20 factor; FST4W-120 the FT factor)
    gen_slot("ft4_fs192000_f24000", "FT4", 192000, 24000, 0xF4, 1440000 // 2048 * 2048, 2048)
    gen_slot("wspr_fs192000_f1500", "WSPR", 192000, 1500, 0x3535, 23040000 // 2048 * 2048, 2048)
    gen_slot("fst4w120_fs192000_f1500", "FST4W-120", 192000, 1500, 0x3535, 23040000 // 2048 * 2048, 2048)
    gen_adversarial()


if __name__ == "__main__":
    main()

// TEST INFRASTRUCTURE ONLY -- never linked into the product path.
//
// Thin extern "C" harness around the *unmodified* reference DSP headers
// (/root/reference/source/SSBD.hpp + LowPass.hpp), compiled in place by
// oracle/Makefile into oracle/_ref/libcwsl_ref.so.  No reference source is
// copied into this repository: the headers are found through -I at build
// time and only exist in the build container.
//
// Build of record: g++ -std=c++17 -O2 -ffp-contract=off,
// no -march=native, no -ffast-math.
//
// The harness exposes the reference's private state (taps, tone, phasor) by
// compiling the header with `private` remapped; this changes access control
// only, not a single arithmetic instruction.
#include <cstddef>
#include <cstdint>
#include <complex>
#include <vector>
#include <stdexcept>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <thread>
#include <chrono>

#include <string>
#include <memory>
#include <atomic>
#include <mutex>
#include <condition_variable>
#include <cctype>

#define private public
#include "SSBD.hpp"
#undef private
// further reference headers that are plain standard C++: the two-deep frame ring and its frames
// (Instance.hpp:95 af_buffer), the slot-clock predicate groups, and the small text helpers of the output stage
#include "ring_buffer.h"
#include "decode_audio_buffer.h"
#include "CWSL_DIGI_Types.hpp"
#include "HamUtils.hpp"
#include "StringUtils.hpp"

extern "C" {

// Opaque handle = SSBD<float>* (the only instantiation the reference uses,
// Instance.cpp:187).
void* ref_ssbd_new(uint64_t Fs, uint64_t B, double F, int usb, char* err, int errlen)
{
    try {
        return new SSBD<float>(Fs, B, F, usb != 0);
    } catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return nullptr;
    }
}

void ref_ssbd_delete(void* h) { delete static_cast<SSBD<float>*>(h); }

uint64_t ref_ssbd_in_size(void* h)  { return static_cast<SSBD<float>*>(h)->GetInSize(); }
uint64_t ref_ssbd_out_size(void* h) { return static_cast<SSBD<float>*>(h)->GetOutSize(); }
uint64_t ref_ssbd_out_rate(void* h) { return static_cast<SSBD<float>*>(h)->GetOutRate(); }
uint64_t ref_ssbd_delay(void* h)    { return static_cast<SSBD<float>*>(h)->GetDelay(); }
uint64_t ref_ssbd_filt_order(void* h) { return static_cast<SSBD<float>*>(h)->FiltOrder; }
uint64_t ref_ssbd_block_size(void* h) { return static_cast<SSBD<float>*>(h)->BlockSize; }

// taps[FiltOrder] (normalised, as the object holds them)
void ref_ssbd_get_taps(void* h, float* taps)
{
    auto* s = static_cast<SSBD<float>*>(h);
    for (size_t n = 0; n < s->FiltOrder; ++n) taps[n] = s->filter[n];
}

// tone[BlockSize] interleaved re,im ; phase_inc (2) ; phase (2)
void ref_ssbd_get_tone(void* h, float* tone_ri, float* phase_inc_ri, float* phase_ri)
{
    auto* s = static_cast<SSBD<float>*>(h);
    for (size_t n = 0; n < s->BlockSize; ++n) {
        tone_ri[2 * n] = s->tone[n].real();
        tone_ri[2 * n + 1] = s->tone[n].imag();
    }
    phase_inc_ri[0] = s->phase_inc.real(); phase_inc_ri[1] = s->phase_inc.imag();
    phase_ri[0] = s->phase.real(); phase_ri[1] = s->phase.imag();
}

// Drive Iterate() exactly as Instance.cpp:273-275 does: n_complex must be a
// multiple of GetInSize(); out receives n_complex / (GetInSize()/4) floats.
// If phase_trace != NULL it receives the phasor (re,im) *before* every block,
// i.e. phase_b for b = 0 .. n_complex/BlockSize-1.
// Returns 0, or -1 if the traced phasor ever disagreed with the object's own
// state after Iterate() (it must not: same operator, same flags).
int ref_ssbd_run(void* h, const float* iq_ri, uint64_t n_complex, float* out, float* phase_trace)
{
    int rc = 0;
    auto* s = static_cast<SSBD<float>*>(h);
    const std::complex<float>* xc = reinterpret_cast<const std::complex<float>*>(iq_ri);
    const size_t in_size = s->GetInSize();
    const size_t dec = in_size / 4;
    for (size_t n = 0; n < n_complex; n += in_size) {
        std::complex<float> p = s->phase;
        if (phase_trace) {
            // phasor before each of the 4 blocks: replay the recurrence on a copy
            for (int k = 0; k < 4; ++k) {
                const size_t b = n / s->BlockSize + k;
                phase_trace[2 * b] = p.real();
                phase_trace[2 * b + 1] = p.imag();
                p *= s->phase_inc;
            }
        }
        s->Iterate(xc + n, out + n / dec);
        if (phase_trace && std::memcmp(&p, &s->phase, sizeof(p)) != 0) rc = -1;
    }
    return rc;
}

// SSBD::Tune on a live object (SSBD.hpp:96-123); 0 ok, -1 = threw (message in err)
int ref_ssbd_tune(void* h, double F, int usb, char* err, int errlen)
{
    try { static_cast<SSBD<float>*>(h)->Tune(F, usb != 0); return 0; }
    catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}
// the same with Tune's third argument (reset = false keeps workspace, index and phase)
int ref_ssbd_tune_ex(void* h, double F, int usb, int reset, char* err, int errlen)
{
    try { static_cast<SSBD<float>*>(h)->Tune(F, usb != 0, reset != 0); return 0; }
    catch (const std::exception& e) {
        if (err && errlen > 0) { std::strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
        return -1;
    }
}

// BuildLowPass<float> alone (LowPass.hpp:16-35), un-normalised.
void ref_build_lowpass(uint64_t order, double bandwidth, float* taps)
{
    float* f = BuildLowPass<float>(order, bandwidth);
    for (size_t n = 0; n < order; ++n) taps[n] = f[n];
    delete[] f;
}

// CPU baseline of kind "reference": the reference's own SSBD<float>::Iterate loop, driven exactly like
// Instance::sampleManager does (one thread per channel, iq_len-sample blocks, a new SSBD per slot; Instance.cpp:251,273-275).
// Returns wall seconds for `threads` channels x `slots` slots of n_per_slot samples, or <0 on error.
double ref_bench_cpu(int threads, int slots, uint64_t Fs, uint32_t iq_len, uint64_t n_per_slot)
{
    if (threads < 1 || threads > 4096 || slots < 1) return -1.0;
    std::vector<std::vector<float>> iq(threads), out(threads);
    for (int t = 0; t < threads; ++t) {
        iq[t].resize(2 * n_per_slot);
        out[t].resize(n_per_slot / (Fs / 12000) + 16);
        uint64_t z = 0x9E3779B97F4A7C15ull * (uint64_t)(t + 1);
        for (size_t k = 0; k < iq[t].size(); ++k) {           // cheap deterministic noise; values do not matter for timing
            z ^= z << 13; z ^= z >> 7; z ^= z << 17;
            iq[t][k] = (float)((int)(z & 0xFFFF) - 32768) * 0.05f;
        }
    }
    std::vector<std::thread> th;
    std::vector<int> rc(threads, 0);
    const auto t0 = std::chrono::steady_clock::now();
    for (int t = 0; t < threads; ++t) {
        th.emplace_back([&, t]() {
            try {
                const std::complex<float>* xc = reinterpret_cast<const std::complex<float>*>(iq[t].data());
                for (int s = 0; s < slots; ++s) {
                    SSBD<float> ssbd(Fs, 6000, (float)(-26000 + 137 * t), true);
                    const size_t in_size = ssbd.GetInSize(), dec = in_size / 4;
                    for (uint64_t b = 0; b + iq_len <= n_per_slot; b += iq_len)
                        for (size_t n = 0; n < iq_len; n += in_size)
                            ssbd.Iterate(xc + b + n, out[t].data() + (b + n) / dec);
                }
            } catch (...) { rc[t] = -1; }
        });
    }
    for (auto& x : th) x.join();
    const auto t1 = std::chrono::steady_clock::now();
    for (int r : rc) if (r) return -1.0;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---- Instance framing replayed on the reference's OWN containers -------------------------------------------------
// Instance.cpp itself needs <windows.h>; its frame handling is three short sequences of calls on ring_buffer_t /
// sample_buffer_t / SSBD, transcribed here call for call (Instance.cpp:139-157 init, :203-227,251 boundary, :268-276
// block).  What the containers and the demodulator DO with those calls is the compiled reference.
struct RefInstance {
    ring_buffer_t<sample_buffer_t<float>> af_buffer;
    std::unique_ptr<SSBD<float>> ssbd;
    uint64_t Fs; uint32_t iq_len; double demodFreq; size_t decRatio, ssbd_in_size;
};

void* ref_inst_new(uint64_t Fs, uint32_t iq_len, double demod_hz, uint64_t frame_len)
{
    try {
        auto* I = new RefInstance();
        I->Fs = Fs; I->iq_len = iq_len; I->demodFreq = demod_hz; I->decRatio = Fs / 12000;
        if (!I->af_buffer.initialize(2)) { delete I; return nullptr; }                      // :143
        for (size_t k = 0; k < I->af_buffer.size; ++k) {                                      // :152-157
            I->af_buffer.recs[k].init(frame_len);
            memset(I->af_buffer.recs[k].buf, 0, I->af_buffer.recs[k].byte_size());
            I->af_buffer.recs[k].resetIndices();
        }
        I->ssbd = std::make_unique<SSBD<float>>(Fs, 6000, static_cast<float>(demod_hz), true);   // :187
        I->ssbd_in_size = I->ssbd->GetInSize();
        return I;
    } catch (...) { return nullptr; }
}
void ref_inst_delete(void* h)
{
    auto* I = static_cast<RefInstance*>(h);
    if (!I) return;
    for (size_t k = 0; k < I->af_buffer.size; ++k) I->af_buffer.recs[k].deallocate();      // frames were malloc'ed by init()
    delete I;
}
// one Receiver block (Instance.cpp:268-276): 1 consumed, 0 "af buffer full"
int ref_inst_push(void* h, const float* iq_ri)
{
    auto* I = static_cast<RefInstance*>(h);
    auto& af = I->af_buffer;
    const std::complex<float>* xc = reinterpret_cast<const std::complex<float>*>(iq_ri);
    if (af.recs[af.write_index].write_index + I->iq_len > af.recs[af.write_index].size - 1) return 0;
    float* dest = af.recs[af.write_index].buf + af.recs[af.write_index].write_index;
    for (size_t n = 0; n < I->iq_len; n += I->ssbd_in_size) I->ssbd->Iterate(xc + n, dest + n / I->decRatio);
    af.recs[af.write_index].write_index += (I->iq_len / I->decRatio);
    return 1;
}
// slot boundary (Instance.cpp:203-227, 251): 1 = frame handed on (copied to frame_out BEFORE prepareAudio), 0 = discarded
int ref_inst_boundary(void* h, uint64_t epoch_s, float* frame_out, uint64_t* t_start, uint64_t* n_written)
{
    auto* I = static_cast<RefInstance*>(h);
    auto& af = I->af_buffer;
    auto idx_next = af.get_next_write_index();
    memset(af.recs[idx_next].buf, 0, af.recs[af.write_index].byte_size());
    af.recs[idx_next].reset();
    af.recs[idx_next].startEpochTime = epoch_s;
    af.inc_write_index();
    auto& current = af.pop_ref();
    const auto startTime = current.startEpochTime;
    if (t_start) *t_start = startTime;
    if (n_written) *n_written = current.write_index;
    if (0 == startTime) return 0;                                                              // :224-227
    if (frame_out) memcpy(frame_out, current.buf, current.byte_size());
    I->ssbd = std::make_unique<SSBD<float>>(I->Fs, 6000, static_cast<float>(I->demodFreq), true);   // :251
    return 1;
}

// slot-clock group of a mode through the reference's SyncPredicates::createPredicate (CWSL_DIGI_Types.hpp:83-134):
// 0 ft8, 1 ft4, 2 q65_30, 3 s60, 4 s120, 5 s300, 6 s900, 7 s1800; -1 = "Unhandled mode"
int ref_mode_group(const char* mode)
{
    try {
        SyncPredicates P;
        auto pred = P.createPredicate(mode);
        const std::vector<std::shared_ptr<SyncPredicate>>* v[8] = {&P.ft8Preds, &P.ft4Preds, &P.q65_30Preds, &P.s60sPreds,
                                                                    &P.s120sPreds, &P.s300sPreds, &P.s900sPreds, &P.s1800sPreds};
        for (int g = 0; g < 8; ++g) if (!v[g]->empty()) return g;
        return -2;
    } catch (const std::exception&) { return -1; }
}

int ref_is_valid_locator(const char* s) { return isValidLocator(std::string(s)) ? 1 : 0; }   // HamUtils.hpp:26-43
void ref_trim(const char* in, char* out, int cap)                                             // StringUtils.hpp:25-28
{
    std::string s(in);
    trim(s);
    std::strncpy(out, s.c_str(), (size_t)cap - 1); out[cap - 1] = 0;
}

} // extern "C"

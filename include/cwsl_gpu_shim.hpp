// cwsl_gpu_shim.hpp -- header-only C++17 shim that turns the C ABI of libcwslgpu.so back into the shapes
// the reference's call sites use, so Receiver / Instance / DecoderPool code compiles against the GPU path
// with one-line changes.
//
//   cwslgpu::Context          process-wide handle (one per GPU)
//   cwslgpu::ReceiverPort     what Receiver::readIQ writes into instead of ring_buffer_spmc_t
//                             (source/Receiver.hpp:247-249)
//   cwslgpu::SsbChannel       SSBD<float>-shaped object: same constructor arguments, same getters (GetInRate, GetOutRate,
//                             GetInSize, GetOutSize, GetBandwidth, GetCarrier, IsUSB, GetDelay: source/SSBD.hpp:140-154),
//                             same Tune, same std::invalid_argument messages (source/SSBD.hpp:48-59,97-103)
//   cwslgpu::FrameSink        what Instance::sampleManager does after a slot boundary (source/Instance.cpp:221-245) for a
//                             set of channels: every newly finalised frame goes to a callback as the ingredients of
//                             ItemToDecode (DecoderPool.hpp:174-210)
//
// SSBD::Iterate(in, out) (SSBD.hpp:127-137) is deliberately absent: its contract is a synchronous 64-samples-in / 4-samples-out
// call, which on a GPU is one kernel launch and one device-to-host wait per 4 output samples; its only caller in the reference
// is the loop of Instance::sampleManager (Instance.cpp:273) that ReceiverPort::push + Context::slotBoundary + SsbChannel::fetch
// replace as a whole.
//
// Nothing here computes DSP: every call forwards to the C ABI.
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include <complex>

#include "cwsl_gpu.h"

namespace cwslgpu {

inline void check(cwslg_ctx *c, int rc)
{
    if (rc >= 0) return;
    const char *detail = c ? cwslg_last_error(c) : "";
    std::string msg = (detail && *detail) ? detail : cwslg_strerror(rc);
    // the reference throws std::invalid_argument for the three tuning errors (SSBD.hpp:54-59,100-103)
    if (rc == CWSLG_ERR_RATIO || rc == CWSLG_ERR_BAND_LOW || rc == CWSLG_ERR_BAND_HIGH)
        throw std::invalid_argument(cwslg_strerror(rc));
    if (rc == CWSLG_ERR_MODE) throw std::runtime_error(msg);          // CWSL_DIGI.hpp:111
    throw std::runtime_error("libcwslgpu: " + msg);
}

class Context {
public:
    explicit Context(int device = -1) { check(nullptr, cwslg_create(&c_, device)); }
    ~Context() { cwslg_destroy(c_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    cwslg_ctx *raw() const { return c_; }
    void setScaleFactors(float ft, float wspr) { check(c_, cwslg_set_scale_factors(c_, ft, wspr)); }
    // SyncPredicate::store(true) for a whole group (CWSL_DIGI.cpp:247-251)
    void slotBoundary(int group, std::uint64_t epoch_s) { check(c_, cwslg_slot_boundary(c_, group, epoch_s)); }
    void process() { check(c_, cwslg_process(c_)); }
    void synchronize() { check(c_, cwslg_synchronize(c_)); }
    // One block for each of several receivers in ONE call (cwslg_push_iq_many): for a host that serves thousands of streams, where a
    // per-receiver push per block (Receiver.hpp:242-249, ReceiverPort::push below) would mean hundreds of thousands of copies a second.
    // ids[k] = ReceiverPort::id(); blocks[k] = that receiver's n_complex samples.
    void pushMany(const std::vector<int> &ids, const std::vector<const std::complex<float> *> &blocks, std::uint32_t n_complex)
    {
        if (ids.size() != blocks.size()) throw std::invalid_argument("pushMany: one block per receiver");
        check(c_, cwslg_push_iq_many(c_, static_cast<int>(ids.size()), ids.data(), reinterpret_cast<const float *const *>(blocks.data()), n_complex));
    }
private:
    cwslg_ctx *c_ = nullptr;
};

// Receiver side: `memcpy(iq_buffer.recs[write_index], rawiq.data(), readSize); iq_buffer.inc_write_index();`
// becomes `port.push(rawiq.data(), iq_len);`
class ReceiverPort {
public:
    ReceiverPort(Context &ctx, std::uint32_t sampleRate, std::uint32_t blockInSamples, std::int32_t lo_hz,
                 std::uint32_t ringBlocks = 0) : ctx_(ctx), fs_(sampleRate)
    {
        check(ctx_.raw(), cwslg_receiver_open(ctx_.raw(), sampleRate, blockInSamples, lo_hz, ringBlocks, &id_));
    }
    ~ReceiverPort() { cwslg_receiver_close(ctx_.raw(), id_); }
    void push(const std::complex<float> *block, std::uint32_t n_complex)
    {
        check(ctx_.raw(), cwslg_push_iq(ctx_.raw(), id_, reinterpret_cast<const float *>(block), n_complex));
    }
    int id() const { return id_; }
    std::uint32_t sampleRate() const { return fs_; }     // Receiver::getSampleRate
    Context &context() const { return ctx_; }
private:
    Context &ctx_;
    int id_ = -1;
    std::uint32_t fs_ = 0;
};

// SSBD<float>(Fs, B, F, isUSB) -> SsbChannel(port, F, isUSB, mode).  Fs comes from the port; B is SSB_BW.
// F is a whole number of Hz: the reference passes static_cast<float>(demodFreq) of an integer FrequencyHz (Instance.cpp:183-187,
// 251), the C ABI takes it as int32 -- a fractional F given here is truncated toward zero, not rounded.
class SsbChannel {
public:
    SsbChannel(ReceiverPort &port, double F, bool isUSB, const std::string &mode)
        : ctx_(port.context()), mode_(mode), fs_(port.sampleRate()), carrier_(static_cast<double>(static_cast<std::int32_t>(F))), usb_(isUSB)
    {
        check(ctx_.raw(), cwslg_channel_open(ctx_.raw(), port.id(), static_cast<std::int32_t>(F), isUSB ? 1 : 0,
                                             mode.c_str(), &id_));
        check(ctx_.raw(), cwslg_channel_info(ctx_.raw(), id_, &in_, &out_, &rate_, &delay_, &frame_));
    }
    ~SsbChannel() { cwslg_channel_close(ctx_.raw(), id_); }
    std::size_t GetInRate() const { return fs_; }       // SSBD.hpp:140
    std::size_t GetOutRate() const { return rate_; }    // :142
    std::size_t GetInSize() const { return in_; }       // :144
    std::size_t GetOutSize() const { return out_; }     // :146
    std::size_t GetBandwidth() const { return rate_ / 2; }   // :148  (GetOutRate() is 2*B)
    double GetCarrier() const { return carrier_; }      // :150  the tuned frequency F as last set (constructor or Tune)
    bool IsUSB() const { return usb_; }                 // :152
    std::size_t GetDelay() const { return delay_; }     // :154
    const std::string &mode() const { return mode_; }
    float trPeriod() const { return static_cast<float>(frame_ / 12000) - 5.0f; }    // frame = 12000 * (period + 5), Instance.cpp:149
    // SSBD::Tune(F, isUSB, reset = true) (SSBD.hpp:97): throws the same std::invalid_argument texts; the old tuning stays in force
    // after a throw.  reset = false keeps filter history, block position and phase, as :116-121 does when skipped.
    void Tune(double F, bool isUSB, bool reset = true)
    {
        const int rc = cwslg_channel_tune_ex(ctx_.raw(), id_, static_cast<std::int32_t>(F), isUSB ? 1 : 0, reset ? 1 : 0);
        if (rc == CWSLG_ERR_BAND_LOW || rc == CWSLG_ERR_BAND_HIGH || rc == CWSLG_ERR_RATIO) throw std::invalid_argument(cwslg_strerror(rc));
        check(ctx_.raw(), rc);
        carrier_ = static_cast<double>(static_cast<std::int32_t>(F));
        usb_ = isUSB;
    }
    std::size_t frameLength() const { return frame_; }  // Instance.cpp:149
    int id() const { return id_; }

    // What Instance.cpp:238-245 hands to DecoderPool::push: returns false while no frame is ready
    // (first partial slot).  audio is resized to frameLength().
    bool fetch(std::vector<std::int16_t> &audio, std::uint64_t &startEpoch)
    {
        audio.resize(frame_);
        std::size_t nv = 0;
        const int rc = cwslg_fetch_frame(ctx_.raw(), id_, audio.data(), audio.size(), &startEpoch, &nv, nullptr);
        if (rc == CWSLG_ERR_NO_FRAME) return false;
        check(ctx_.raw(), rc);
        return true;
    }
    // startEpoch (optional): the start epoch of the frame the list was computed from -- pair it with fetch()'s, or use fetchSlot()
    int candidates(std::vector<cwslg_candidate> &out, int max = 600, std::uint64_t *startEpoch = nullptr)
    {
        out.resize(max);
        int n = 0;
        check(ctx_.raw(), cwslg_fetch_candidates(ctx_.raw(), id_, out.data(), max, &n, startEpoch));
        out.resize(n);
        return n;
    }
    // The ItemToDecode of one slot (DecoderPool.hpp:174-210: audio + epochTime travel together) with the candidate list of the SAME epoch,
    // in one call and under one ticket (cwslg_fetch_slot): false while no frame is ready.  FT8 / FT4 channels.
    bool fetchSlot(std::vector<std::int16_t> &audio, std::uint64_t &startEpoch, std::vector<cwslg_candidate> &cands, int max = 600)
    {
        audio.resize(frame_);
        cands.resize(max);
        cwslg_slot_result r;
        const int rc = cwslg_fetch_slot(ctx_.raw(), id_, audio.data(), audio.size(), cands.data(), cands.size() * sizeof(cwslg_candidate),
                                        nullptr, 0, &r);
        if (rc == CWSLG_ERR_NO_FRAME) { cands.clear(); return false; }
        check(ctx_.raw(), rc);
        startEpoch = r.start_epoch;
        cands.resize((r.list_kind == CWSLG_LIST_FT8 || r.list_kind == CWSLG_LIST_FT4) ? r.n_list : 0);
        return true;
    }
    // FT4 channels: every candidate refined coherently (start sample, frequency tweak, sync) -- cwslg_fetch_ft4_sync
    int ft4Sync(std::vector<cwslg_ft4_sync> &out, int max = 1800)
    {
        out.resize(max);
        int n = 0;
        const int rc = cwslg_fetch_ft4_sync(ctx_.raw(), id_, out.data(), max, &n, nullptr);
        if (rc == CWSLG_ERR_NO_FRAME) { out.clear(); return 0; }
        check(ctx_.raw(), rc);
        out.resize(n);
        return n;
    }
    // what decodeUsingShMem does between lock and unlock (DecoderPool.hpp:451-590): the jt9 block, d2 straight from HBM
    bool fillDecoderBlock(void *block, std::size_t bytes, int decodedepth, int highestDecodeFreq, std::uint64_t &startEpoch, bool js8 = false)
    {
        const int rc = cwslg_fill_decoder_block(ctx_.raw(), id_, block, bytes, js8 ? 1 : 0, decodedepth, highestDecodeFreq, &startEpoch);
        if (rc == CWSLG_ERR_NO_FRAME) return false;
        check(ctx_.raw(), rc);
        return true;
    }
private:
    Context &ctx_;
    int id_ = -1;
    std::string mode_;
    std::uint32_t fs_ = 0;
    double carrier_ = 0.0;
    bool usb_ = true;
    std::uint32_t in_ = 0, out_ = 0, rate_ = 0, delay_ = 0;
    std::size_t frame_ = 0;
};

// What Instance::sampleManager does with a finished frame (Instance.cpp:221-245), for every channel registered here:
//     ItemToDecode toDecode(audioBuf_i16, digitalMode, startTime, ssbFreq, static_cast<int>(id), cwd, trperiod);
//     decoderPool->push(toDecode);
// Call collect() after Context::slotBoundary(): each channel whose last finalised frame is new (its startEpochTime differs from the one
// delivered before; the first, discarded slot yields none, :224-227) is fetched and handed to the callback with those arguments.
class FrameSink {
public:
    using Callback = std::function<void(std::vector<std::int16_t> &&audio, const std::string &mode, std::uint64_t epochTime,
                                        std::int64_t baseFreq, int instanceId, const std::string &cwd, float trperiod)>;
    // ABI 5: the same hand-off with the candidate list of the SAME epoch beside the audio (cwslg_fetch_slot: one call, one ticket) -- for a decoder
    // pool whose ItemToDecode (DecoderPool.hpp:174-210) is extended by a candidate vector; FT8 / FT4 channels with the sync stage on, empty otherwise
    using SlotCallback = std::function<void(std::vector<std::int16_t> &&audio, const std::string &mode, std::uint64_t epochTime,
                                            std::int64_t baseFreq, int instanceId, const std::string &cwd, float trperiod,
                                            std::vector<cwslg_candidate> &&candidates)>;
    explicit FrameSink(Callback cb) : cb_(std::move(cb)) {}
    explicit FrameSink(SlotCallback cb, int maxCandidates = 600) : scb_(std::move(cb)), max_cand_(maxCandidates) {}
    // ssbFreq, instanceId, cwd: Instance's members of the same names (Instance.cpp:121-176)
    void add(SsbChannel &ch, std::int64_t ssbFreq, int instanceId, const std::string &cwd)
    {
        entries_.push_back(Entry{&ch, ssbFreq, instanceId, cwd, 0, false});
    }
    // returns the number of frames delivered
    int collect()
    {
        int n = 0;
        for (Entry &e : entries_) {
            std::vector<std::int16_t> audio;
            std::vector<cwslg_candidate> cands;
            std::uint64_t t0 = 0;
            if (!(scb_ ? e.ch->fetchSlot(audio, t0, cands, max_cand_) : e.ch->fetch(audio, t0))) continue;
            if (e.seen && t0 == e.last) continue;            // that frame went out at an earlier boundary
            e.seen = true;
            e.last = t0;
            if (scb_) scb_(std::move(audio), e.ch->mode(), t0, e.ssbFreq, e.instanceId, e.cwd, e.ch->trPeriod(), std::move(cands));
            else cb_(std::move(audio), e.ch->mode(), t0, e.ssbFreq, e.instanceId, e.cwd, e.ch->trPeriod());
            ++n;
        }
        return n;
    }
private:
    struct Entry { SsbChannel *ch; std::int64_t ssbFreq; int instanceId; std::string cwd; std::uint64_t last; bool seen; };
    Callback cb_;
    SlotCallback scb_;
    int max_cand_ = 600;
    std::vector<Entry> entries_;
};

} // namespace cwslgpu

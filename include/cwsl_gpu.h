/*
 * cwsl_gpu.h -- C ABI of libcwslgpu.so, the MI355X (gfx950) replacement for the
 * per-Instance CPU DSP chain of CWSL_DIGI.
 *
 * The reference has no plugin/FFI layer: the seam is a handful of C++ call sites.
 * Every entry point below names the reference interface it replaces (paths are
 * relative to the reference's source/ directory).  include/cwsl_gpu_shim.hpp
 * wraps this ABI back into SSBD- and Instance-shaped C++ classes so those call
 * sites compile unchanged; INTEGRATION.md shows the binding.
 *
 * Conventions: plain pointers and sizes only; every function returns a CWSLG_*
 * status (0 = OK, <0 = error); the library never falls back to a CPU path --
 * without a usable HIP device cwslg_create() fails with CWSLG_ERR_NO_DEVICE.
 * All entry points are thread-safe with respect to one context (one internal
 * mutex; device work is serialised on the context's HIP stream), mirroring the
 * reference's {Receiver thread} || {slot-clock thread} || {consumer} concurrency.
 */
#ifndef CWSL_GPU_H
#define CWSL_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CWSLG_ABI_VERSION 5

/* ---- status codes ---- */
#define CWSLG_OK                  0
#define CWSLG_ERR_RATIO          -1  /* SSBD.hpp:54-55   "Fs/B must be an even integer >= 4"      */
#define CWSLG_ERR_BAND_LOW       -2  /* SSBD.hpp:100-101 "Signal outside of band (low)"           */
#define CWSLG_ERR_BAND_HIGH      -3  /* SSBD.hpp:102-103 "Signal outside of band (high)"          */
#define CWSLG_ERR_NOMEM          -4
#define CWSLG_ERR_MODE           -5  /* CWSL_DIGI.hpp:110-112 "Unhandled mode: ..."               */
#define CWSLG_ERR_ARG            -6
#define CWSLG_ERR_NO_DEVICE      -7  /* no HIP device / wrong arch: the product never runs on CPU */
#define CWSLG_ERR_HIP            -8
#define CWSLG_ERR_NO_FRAME       -9  /* nothing to fetch (e.g. first partial slot, Instance.cpp:224-227) */
#define CWSLG_ERR_UNSUPPORTED   -10  /* sample rate other than CWSL's 48/96/192 kHz               */
#define CWSLG_ERR_BLOCK         -11  /* n_complex not a multiple of SSBD::GetInSize() (= 2*Fs/B)  */

/* ---- slot-clock groups: CWSL_DIGI_Types.hpp:83-143 (SyncPredicates vectors) ---- */
#define CWSLG_GROUP_FT8     0   /* FT8, JS8                       (ft8Preds)    */
#define CWSLG_GROUP_FT4     1   /* FT4                            (ft4Preds)    */
#define CWSLG_GROUP_Q65_30  2   /* Q65-30                         (q65_30Preds) */
#define CWSLG_GROUP_S60     3   /* JT65, FST4-60                  (s60sPreds)   */
#define CWSLG_GROUP_S120    4   /* WSPR, FST4-120, FST4W-120      (s120sPreds)  */
#define CWSLG_GROUP_S300    5   /* FST4-300, FST4W-300            (s300sPreds)  */
#define CWSLG_GROUP_S900    6   /* FST4-900, FST4W-900            (s900sPreds)  */
#define CWSLG_GROUP_S1800   7   /* FST4-1800, FST4W-1800          (s1800sPreds) */
#define CWSLG_NUM_GROUPS    8

typedef struct cwslg_ctx cwslg_ctx;

/* One sync candidate (row a13 of SURVEY.md 8a: no reference counterpart; the layout is
 * builder-defined and the ordering is deterministic: by default descending sync, ties by ascending
 * freq_bin then ascending time_step; see cwslg_set_candidate_order). */
typedef struct {
    int32_t freq_bin;     /* FT8: 3.125 Hz bins of the 3840-point symbol spectrum               */
    int32_t time_step;    /* FT8: lag in quarter-symbol steps (40 ms), -62..+62                 */
    float   sync;         /* normalised Costas sync metric                                      */
    float   freq_hz;      /* freq_bin * df                                                      */
    float   dt_s;         /* (time_step - 0.5 s offset convention of the upstream decoder)      */
} cwslg_candidate;

/* ---- context ---- */
int  cwslg_abi_version(void);
/* device_ordinal < 0: use LOCAL_RANK from the environment if set, else 0. */
int  cwslg_create(cwslg_ctx **out, int device_ordinal);
void cwslg_destroy(cwslg_ctx *ctx);
const char *cwslg_strerror(int status);
/* Text of the last failure on this context (HIP error string included).  Safe from any thread; the pointer is to a per-thread copy,
 * valid until the same thread calls this function again. */
const char *cwslg_last_error(cwslg_ctx *ctx);
/* wsjtx.ftaudioscalefactor / wsjtx.wspraudioscalefactor (CWSL_DIGI.cpp:100-101; defaults 0.90 / 0.20) */
int  cwslg_set_scale_factors(cwslg_ctx *ctx, float scale_ft, float scale_wspr);

/* Arithmetic mode of the demodulator.
 *   on = 1, the DEFAULT of a new context ("exact"): the reference's own operation order (SSBD.hpp:160-183, un-fused float32,
 *           tap blocks oldest first).  The float frame, the int16 frame and therefore every candidate list equal the reference
 *           chain's (SSBD + Instance::prepareAudio + int16 conversion compiled from the reference's headers) BIT FOR BIT.
 *   on = 0  ("fast"): the fused polyphase form of the same filter (FMA, different summation order): about 1.5x the throughput of the whole path;
 *           float audio within 1e-5 of the frame's peak (measured 4e-7 with in-band signals; up to 6.6e-6 when the band is empty next
 *           to a strong out-of-band carrier, because the rounding noise scales with the INPUT level while the bound is relative to
 *           the frame's own peak -- DESIGN.md section 2 derives the bound); int16 samples differ by at most 1 LSB, and only where
 *           x*factor + 0.5 lies within that error of an integer; candidate lists equal the reference chain's keys except within 1e-3
 *           of a threshold, sync values within 1e-3.
 * Samples already pushed keep the mode they were pushed under.  A real-time receiver needs 0.3 % of either mode's throughput. */
int  cwslg_set_exact(cwslg_ctx *ctx, int on);

/* ---- receivers: replaces Receiver::init + the SPMC ring (Receiver.hpp:115-163, ring_buffer_spmc.h) ----
 * fs, iq_len, lo_hz are SM_HDR.SampleRate / BlockInSamples / L0 (SharedMemory.h:10-21, Receiver.hpp:86-88).
 * ring_blocks = 0 selects the reference depth 3*(fs/iq_len+1) blocks (Receiver.hpp:132).  The ring lives in HBM and is shorter than 4 GiB
 * (512 M samples: CWSLG_ERR_ARG beyond). */
int cwslg_receiver_open(cwslg_ctx *ctx, uint32_t fs, uint32_t iq_len, int32_t lo_hz,
                        uint32_t ring_blocks, int *rx_id);
int cwslg_receiver_close(cwslg_ctx *ctx, int rx_id);
/* Replaces the memcpy into the ring slot + inc_write_index (Receiver.hpp:247-249) AND the N per-Instance
 * pop_no_wait()/Iterate() loops (Instance.cpp:265-276): called ONCE per block per receiver.  The host block
 * is only read during the call.  n_complex must be a multiple of 2*fs/6000 (SSBD::GetInSize).
 * One call is one H2D copy and two context-mutex acquisitions: right for the reference's topology (a few receivers, many channels each).
 * A host that feeds THOUSANDS of private streams in real time must use cwslg_push_iq_many below (measured, profiles/r4_realtime.json: 4096
 * streams through this call fall behind -- 13.1 s of wall time for 5.2 s of signal; through the batched call 0 drops at 6.3 GB/s on a
 * quarter of a host core).  A receiver has one pusher at a time (Receiver.hpp:167); a second concurrent push of the same receiver,
 * by either call, is refused with CWSLG_ERR_ARG and nothing is accounted. */
int cwslg_push_iq(cwslg_ctx *ctx, int rx_id, const float *iq_interleaved, uint32_t n_complex);
/* One block of n_complex samples for EACH of n_rx receivers in one call -- the batched form of the per-block memcpy + inc_write_index of
 * Receiver::readIQ (Receiver.hpp:242-249) for a host that serves thousands of streams (the north star's 4096 private 192 kHz streams are
 * 384 000 blocks a second: through cwslg_push_iq that is as many H2D copies and twice as many context-mutex acquisitions; through this
 * call, with one batch per block period, 94 launches and 188 acquisitions).  iq[k] is receiver rx_ids[k]'s block (interleaved re, im; read
 * only during the call); every receiver of a batch gets the same n_complex (a multiple of SSBD::GetInSize for each of them) and may appear
 * once.  The blocks are copied into one pinned, host-mapped staging buffer (sized on demand) and ONE kernel scatters them into the rings;
 * frame-overflow accounting is per receiver with its own iq_len, exactly as n_rx calls of cwslg_push_iq would do it.  Thread safety: several
 * threads may push disjoint batches concurrently; a receiver must not be pushed from two threads at once (the reference has one thread per
 * Receiver, Receiver.hpp:167) -- detected and refused with CWSLG_ERR_ARG, nothing accounted. */
int cwslg_push_iq_many(cwslg_ctx *ctx, int n_rx, const int *rx_ids, const float *const *iq_interleaved, uint32_t n_complex);
/* Same, source already in device memory (tests, device-side producers). */
int cwslg_push_iq_device(cwslg_ctx *ctx, int rx_id, const void *d_iq_interleaved, uint32_t n_complex);
/* Synthetic IQ source standing in for CW Skimmer's shared memory (SharedMemory.cpp is Win32-only):
 * appends n_complex samples generated on the device.  The generator is the portable one specified in
 * DESIGN.md (integer Irwin-Hall noise + table-lookup tones) and is bit-identical to the oracle's.
 * tones_hz are relative to the LO; block_len = the push granularity used for the frame-overflow guard. */
int cwslg_push_synth(cwslg_ctx *ctx, int rx_id, uint64_t seed, uint32_t n_complex, uint32_t block_len,
                     const double *tones_hz, int n_tones, float amp);

/* Zero-copy producer: declare that the next n_complex samples ALREADY in the ring (written by a device-side
 * producer, or left there by an earlier lap) are new input.  Host bookkeeping only; nothing is copied.
 * block_len = granularity of the frame-overflow guard (0 = the receiver's iq_len). */
int cwslg_ring_commit(cwslg_ctx *ctx, int rx_id, uint32_t n_complex, uint32_t block_len);
/* cwslg_ring_commit on every open receiver of the context (one call per tick for thousands of streams). */
int cwslg_ring_commit_all(cwslg_ctx *ctx, uint32_t n_complex, uint32_t block_len);
/* Device address and capacity (complex samples) of the ring; *write_pos = ring index of the next sample. */
int cwslg_ring_info(cwslg_ctx *ctx, int rx_id, void **d_ring, uint32_t *capacity, uint64_t *total_pushed);

/* ---- channels: replaces SSBD<float>(Fs, SSB_BW, (float)demodFreq, USB) + Instance::init
 *      (Instance.cpp:121-176,183-187).  demod_hz = calibratedSSBFreq - LO (Instance.cpp:183).
 *      mode is the decoder= line's mode string ("FT8", "FT4", "WSPR", "FST4W-120", ...). ---- */
int cwslg_channel_open(cwslg_ctx *ctx, int rx_id, int32_t demod_hz, int usb, const char *mode, int *ch_id);
int cwslg_channel_close(cwslg_ctx *ctx, int ch_id);
/* SSBD::Tune(F, isUSB) with its default reset = true (SSBD.hpp:96-123): the channel gets a new tone and phasor step, its
 * filter history is forgotten and its phasor restarts at (1, 0); samples pushed before the call keep the old tuning and
 * the frame keeps filling where it was.  The band checks fail with the same two statuses / messages as at open, and then
 * leave the old tuning in place.  Like a Tune() on Instance's live SSBD object, the retune lasts until the demodulator is next
 * re-created: after every emitted frame Instance constructs a new SSBD from its OWN demodFreq / USB (Instance.cpp:251), and so
 * does cwslg_slot_boundary -- the channel is back on the tuning it was opened with.  (A lasting change of frequency is a new
 * channel, as band rotation is a new Instance in the reference, CWSL_DIGI.cpp:1217-1226.) */
int cwslg_channel_tune(cwslg_ctx *ctx, int ch_id, int32_t demod_hz, int usb);
/* SSBD::Tune(F, isUSB, reset) with its third argument (SSBD.hpp:97, :116-121).  reset != 0 is cwslg_channel_tune.  reset == 0 is the
 * phase-continuous retune: the filter history stays as it was mixed with the old tuning, the phasor continues from its live value
 * with the new step, the block count (Iterate's position) goes on -- so the 31 outputs after the retune point blend the two
 * tunings exactly as the reference's workspace does (bit-identical in exact mode; in the default mode those outputs are computed
 * in the reference's order and the rest within the usual 1e-5).  The reference never passes reset = false itself.
 * A second reset == 0 retune before 32 more blocks have been pushed returns CWSLG_ERR_UNSUPPORTED. */
int cwslg_channel_tune_ex(cwslg_ctx *ctx, int ch_id, int32_t demod_hz, int usb, int reset);
/* SSBD getters (SSBD.hpp:140-154) for the shim */
int cwslg_channel_info(cwslg_ctx *ctx, int ch_id, uint32_t *in_size, uint32_t *out_size,
                       uint32_t *out_rate, uint32_t *delay, size_t *frame_len);

/* ---- config.ini `decoder=` lines (source/CWSL_DIGI.cpp:731-837), accepted unchanged ----
 * "<freqHz> <MODE> [smnum [freqcal [callsign]]]" split on single spaces exactly like splitStringByDelim
 * (StringUtils.hpp:48-66: consecutive spaces yield empty fields), 2..5 fields, callsign only for WSPR.
 * calibrated_hz = (uint32)(freq / (freqcal_global * freqcal)) (:834).  No context needed. */
typedef struct {
    uint32_t freq_hz;          /* dial frequency as written                                   */
    uint32_t calibrated_hz;    /* what Instance tunes to (before subtracting the LO)          */
    char     mode[16];
    int32_t  smnum;            /* shared-memory interface number, -1 if not given (use radio.sharedmem) */
    double   freqcal;          /* per-decoder calibration factor (default 1.0)                */
    char     callsign[16];     /* per-decoder callsign (WSPR only), "" if not given           */
    int32_t  group;            /* CWSLG_GROUP_* of the mode                                   */
    uint32_t frame_len;        /* 12000*(period+5)                                            */
    float    period_s;
} cwslg_decoder_spec;
int cwslg_parse_decoder_line(const char *line, double freqcal_global, cwslg_decoder_spec *out);
/* Parse + open in one step: demod_hz = calibrated_hz - lo_hz of the receiver (Instance.cpp:183), USB. */
int cwslg_channel_open_line(cwslg_ctx *ctx, int rx_id, const char *line, double freqcal_global, int *ch_id);

/* ---- processing ----
 * Demodulate everything pushed so far for every channel (enqueued on the context stream; returns
 * without waiting).  push/slot_boundary/fetch call it implicitly when they have to. */
int cwslg_process(cwslg_ctx *ctx);
/* When is a cwslg_process() call worth a launch (ABI 5)?  The reference wakes every Instance once per block (Instance.cpp:260-276); a host
 * that mirrors this calls cwslg_process() after every push.  The bit-identical kernel starts every stream of outputs 32 blocks early (the
 * workspace of SSBD::ProcessBlock needs an output's 32 blocks), so a launch over 128 pending outputs per channel fetches and multiplies several
 * times what it delivers.  min_outputs = 0 (default): every call launches, as documented above.  min_outputs > 0: a call returns at once until
 * some channel has that many 12 kHz outputs pending.  min_outputs < 0: the library's own threshold (exact mode 20480 outputs = 1.7 s: one wave
 * per channel, 32 streams of 640 outputs, 5 % warm-up; fast mode 2048).  A slot boundary, a retune, a mode switch and ring pressure
 * (Receiver.hpp:222-229, "I/Q buffer is full!") always demodulate everything pending, whatever the threshold -- no sample is ever late for
 * its frame.  stats.demod_blocks_read x D / stats.demod_samples is the redundancy actually paid; stats.process_deferred counts the calls
 * that returned without a launch. */
int cwslg_set_process_threshold(cwslg_ctx *ctx, int min_outputs);
/* Demodulate everything pending NOW, whatever the threshold says (enqueued; returns without waiting).  For a host with a slot clock: called
 * ~100 ms before a boundary it leaves the boundary only the last few blocks to demodulate, so that frames and lists are final one
 * sync stage after the boundary call instead of one demod launch + one sync stage (cwsl_gpu_realtime --flush-before: 4096 channels,
 * 9.9-10.2 instead of 13.6 ms, profiles/r6_realtime.json). */
int cwslg_flush(cwslg_ctx *ctx);
/* The launch policy of the bit-identical kernel as a pure function (no context needed; for tests and capacity planning): outputs per stream for a
 * launch over total_blocks pending blocks (one block = one 12 kHz output), max_blocks of them on the busiest channel, on a chip of cu_count CUs
 * (0 = 256).  Every stream pays a 32-block warm-up, so a launch fetches and multiplies (1 + 32 / length) x what it delivers.  latency != 0: a
 * boundary's rule (the shortest launch: total / 65536, at least 4); 0: the rule of cwslg_process / cwslg_flush / ring pressure (at least one
 * full wave per channel: max_blocks / 32).  At most 1408. */
unsigned cwslg_exact_stream_length(uint64_t total_blocks, unsigned max_blocks, unsigned cu_count, int latency);
/* Replaces SyncPredicate::store(true) for every predicate of one group (CWSL_DIGI.cpp:247-251) and the
 * per-Instance reaction to it (Instance.cpp:203-253): swap frames, stamp the new frame with epoch_s,
 * finalise (peak-normalise + int16) the finished one unless its start time is 0, restart the demodulator. */
int cwslg_slot_boundary(cwslg_ctx *ctx, int group, uint64_t epoch_s);
/* (cwslg_slot_boundary and _begin fail with CWSLG_ERR_ARG while a boundary opened by _begin has not been ended.)
 * The same boundary in two halves, for a throughput host that keeps the GPU busy across boundaries (bench.py with N > 1): _begin queues
 * the boundary's device work and returns; the host queues the next slot's demodulation; _end waits for the boundary's own kernels
 * (not for the stream) and runs the rendezvous, which thereby overlaps the next demod launch.  Without a rendezvous installed _end
 * does nothing.  Do not fetch the epoch's frames or candidates before _end has returned; one boundary may be open at a time. */
int cwslg_slot_boundary_begin(cwslg_ctx *ctx, int group, uint64_t epoch_s);
int cwslg_slot_boundary_end(cwslg_ctx *ctx);
/* Same for a single channel (one SyncPredicate). */
int cwslg_slot_boundary_channel(cwslg_ctx *ctx, int ch_id, uint64_t epoch_s);
int cwslg_synchronize(cwslg_ctx *ctx);

/* ---- multi-GPU: the slot-boundary rendezvous ----
 * Channels never exchange data (Instance.cpp:260-276 reads its receiver's IQ and nothing else), so decoders shard over
 * one process per GPU -- by receiver first, as CWSL_DIGI.cpp:115-129 creates one Receiver per band -- with no data-path
 * collective.  What the reference's shared SyncPredicates give it for free (CWSL_DIGI_Types.hpp:83-143: every Instance
 * of a period family sees the same store(true)) becomes one tiny collective here: when a rendezvous is installed,
 * cwslg_slot_boundary() -- after it has queued finalise (+ sync) for its own channels and waited for its stream --
 * calls it with the number of frames this process emitted for the epoch and receives the total over all processes, so
 * every GPU publishes the epoch together.  The callback runs on the calling thread with the context unlocked; it must
 * not call back into the same context.  Return 0 or a negative CWSLG_ERR_*; the total is kept in the stats. */
typedef int (*cwslg_rendezvous_fn)(void *user, int group, uint64_t epoch_s, uint64_t frames_local, uint64_t *frames_total);
int cwslg_set_boundary_rendezvous(cwslg_ctx *ctx, cwslg_rendezvous_fn fn, void *user);
/* Built-in rendezvous (C/C++ hosts; bench.py's default for N > 1): ONE all-gather of 32 bytes per rank on RCCL over xGMI -- (frames,
 * group, epoch, flag) -- enqueued on the context's SIDE stream (the context stream may already hold the next slot's demodulation): every
 * rank sums the frames and checks that all ranks are at the same group and epoch; a mismatch fails the boundary on every rank with
 * CWSLG_ERR_ARG.  librccl is opened on first use (a process that already carries one, e.g. torch's, shares it).
 * cwslg_rccl_unique_id fills the 128-byte ncclUniqueId on rank 0; the host program hands it to the other ranks by any
 * means (cwsl_gpu_skimmer: a file); every rank then calls cwslg_rccl_init, which installs the rendezvous. */
/* A 64-bit flag word this process contributes to every later built-in rendezvous; stats.rendezvous_flags_and is the AND over all ranks
 * at the last one.  cwsl_gpu_skimmer sets bit 0 when its inputs are exhausted and leaves its loop when every rank has: a rank whose
 * band ends early keeps making the (collective) boundary calls until then instead of leaving the others blocked.  Only the built-in
 * form carries it (a callback rendezvous sees this process's own value). */
int cwslg_set_rendezvous_flag(cwslg_ctx *ctx, uint64_t flag);
#define CWSLG_RCCL_ID_BYTES 128
int cwslg_rccl_unique_id(void *id_out);
int cwslg_rccl_init(cwslg_ctx *ctx, const void *id, int rank, int world);

/* ---- results: replaces decoderPool->push(ItemToDecode(audio_i16, ...)) (Instance.cpp:244-245) ----
 * Copies the last finalised frame of the channel: frame_len = 12000*(period+5) int16 samples with the
 * reference's zero tail; *n_valid = samples actually demodulated in the slot; *start_epoch = the frame's
 * startEpochTime.  Returns CWSLG_ERR_NO_FRAME until a frame has been finalised.
 * Lifetime: the frame stays fetchable until the channel's NEXT emitting boundary replaces it.  The copy is atomic against that boundary
 * (the reference copies the frame into the ItemToDecode it pushes, Instance.cpp:238-245, DecoderPool.hpp:174-210): a fetch that started
 * before the boundary returns the old frame whole with the old start_epoch -- the boundary waits for it -- and one that starts after it
 * returns the new frame with the new start_epoch; samples of one slot are never delivered under the epoch of another.  A consumer more
 * than one slot late therefore sees only the newest frame (the reference's pool would drop the stale item by age, DecoderPool.hpp:358-374). */
int cwslg_fetch_frame(cwslg_ctx *ctx, int ch_id, int16_t *dst, size_t cap,
                      uint64_t *start_epoch, size_t *n_valid, float *factor);
/* Fetches and boundaries (ABI 5).  Every fetch of a result -- cwslg_fetch_frame, cwslg_fetch_slot, cwslg_fill_decoder_block and the four
 * candidate-list fetches -- reads only pointers under the context mutex, takes a ticket of the channel's slot-clock GROUP and copies with the
 * mutex released, on a fetch stream behind an event recorded after the boundary's kernels.  The next boundary of THAT group is the only writer
 * of those buffers and waits for the group's tickets before it queues its kernels, holding the context mutex.  Stall bound: a boundary of group
 * g can hold the context (pushes, cwslg_process, stats) for at most the longest copy of group g still in flight -- one frame + lists, i.e.
 * 0.36 MB (FT8) ... 3 MB (120 s modes) at PCIe speed, plus, for a fetch issued while kernels were still queued, the wait for those kernels
 * (at most the previous boundary's finalise + sync, or a demod launch: <= 40 ms at 4096 slots, ~0.1 ms in real-time operation).  Fetches of
 * OTHER groups never delay it.  A consumer that fetches within its slot period (the reference's pool drops items older than that,
 * DecoderPool.hpp:358-374) never meets the wait at all. */
#define CWSLG_LIST_NONE   0
#define CWSLG_LIST_FT8    1   /* cwslg_candidate records                                            */
#define CWSLG_LIST_FT4    2   /* cwslg_candidate records (+ cwslg_ft4_sync refinements)             */
#define CWSLG_LIST_WSPR   3   /* cwslg_wspr_candidate records                                       */
#define CWSLG_LIST_FST4W  4   /* cwslg_fst4w_candidate records                                      */
typedef struct {
    uint64_t start_epoch;     /* the frame's startEpochTime (Instance.cpp:215); every list below was computed from THIS frame */
    uint64_t n_valid;         /* samples demodulated in the slot                                                            */
    float    factor;          /* prepareAudio's scale factor                                                                */
    int32_t  list_kind;       /* CWSLG_LIST_*: record type written to `list`; NONE if no list of this epoch exists           */
    int32_t  n_list;          /* records written to `list`                                                                  */
    int32_t  n_ft4_sync;      /* records written to `ft4` (FT4 channels with the coherent stage on)                          */
} cwslg_slot_result;
/* The results of ONE epoch of one channel in one call and under one ticket -- the GPU-side counterpart of the ItemToDecode that carries
 * audio + startEpochTime together (DecoderPool.hpp:174-210, Instance.cpp:238-245): the int16 frame (frame may be NULL), the scale factor,
 * the channel's candidate list in its own record type (list_bytes = capacity of `list` in bytes; 0 / NULL to skip) and, for FT4 channels, the
 * coherent refinements.  Frame and lists can never belong to different slots. */
/* The last finalised frame as the reference's .wav (WaveFile.hpp:19-35,87-135: 46-byte RIFF/WAVE/fmt(18-byte
 * WAVEFORMATEX, PCM, mono, 12 kHz, 16 bit)/data header + the whole int16 frame) -- the file jt9/wsprd are given in
 * transfermethod=wavefile mode (DecoderPool.hpp:966-1046).  For the shared-memory mode pass &dec_data->d2[0]
 * to cwslg_fetch_frame instead (DecoderPool.hpp:588). */
int cwslg_write_wav(cwslg_ctx *ctx, int ch_id, const char *path);
/* ---- host service pieces: pure functions, usable without a context -----------------------
 * Slot clock: the first boundary instant (UTC milliseconds) of a mode group strictly after `after_ms` -- what the
 * reference's waitForTime* polling threads (CWSL_DIGI.cpp:174-451) converge on within one 25 ms poll: FT8 :00/:15/
 * :30/:45; FT4 the same plus 7.4 s later (the thread sleeps to 400 ms into seconds 7/22/37/52, :427-437); Q65-30
 * :00/:30; 60 s; 120/300/900/1800 s at second 0 of minutes divisible by 2/5/15/30.  Returns 0 for a bad group.
 * The epoch to hand to cwslg_slot_boundary() is edge_ms / 1000 (Instance.cpp:214 stamps whole seconds). */
uint64_t cwslg_slot_clock_next(int group, uint64_t after_ms);
/* Decoder-pool sizing (CWSL_DIGI.cpp:857-887).  counts[8] = decoders of FT4, FT8, Q65-30, JS8, WSPR, JT65, FST4W-*,
 * FST4-* in that order. */
int cwslg_pool_sizing(const int *counts, float decoderburden, int n_decoders, int *numjt9instances, int *maxwsprdinstances);
/* findBand (CWSL_Utils.hpp:28-55): index of the first band with |f - L0| <= Fs/2, or -1. */
int cwslg_find_band(const int64_t *lo_hz, const uint32_t *fs_hz, int n_bands, int64_t f_hz);

/* ---- decoder stdout -> spot record, every mode but JS8 -----------------------------------------------
 * JT65: "HHMM snr  dt freq #  message" (OutputHandler.cpp:623-695); Q65-30: FT8's columns (:697-780);
 * FST4-*: "HHMM snr  dt freq `  message" (OutputHandler.cpp:243-312); FST4W-*: the same columns with call, locator and
 * dBm as tokens (:152-240); WSPR: wsprd's eight tokens "id snr dt MHz drift call locator dBm" (:314-402).
 * One line of jt9's stdout ("HHMMSS snr  dt freq ~  message", fixed columns: OutputHandler.cpp:505-621) -> the
 * arguments reporter->handle() would receive (message rules: OutputHandler.cpp:924-1128; call / locator checks
 * :788-922, HamUtils.hpp:26-43).  base_freq_hz = the decoder's dial frequency (ItemToDecode::baseFreq).
 * CWSLG_SPOT_OK: call (and locator if has_locator) identify the transmitting station; CWSLG_SPOT_UNHANDLED: a well
 * formed line whose message the reference logs as "Message not handled"; CWSLG_SPOT_SKIP: not a decode line. */
#define CWSLG_SPOT_OK         0
#define CWSLG_SPOT_UNHANDLED  1
#define CWSLG_SPOT_SKIP       2
typedef struct {
    int32_t  snr_db;
    float    dt_s;
    uint32_t freq_hz;          /* audio offset + base frequency, truncated like static_cast<uint32_t> */
    int32_t  has_locator;
    char     call[16];
    char     locator[8];
    char     message[64];      /* the message text, trimmed (FT8 / FT4 / FST4)                        */
    int32_t  drift;            /* WSPR: Hz per minute                                                 */
    int32_t  dbm;              /* WSPR / FST4W: reported power                                        */
} cwslg_spot;
int cwslg_parse_decode_line(const char *mode, const char *line, int64_t base_freq_hz, cwslg_spot *out);

/* ---- decoder hand-off formats -------------------------------------------------------------
 * The block a stock jt9 (js8 = 0: dec_data_t, DecoderPool.hpp:58-108, "in sync with lib/jt9com.f90") or js8
 * (js8 = 1: dec_data_js8_t, :110-171) maps as shared memory.  cwslg_fill_decoder_block() does what
 * decodeUsingShMem() does between lock and unlock (:451-590): zero the block, set params and ipc for the channel's
 * mode, copy min(frame, 30*60*12000) int16 samples into d2 -- here straight from HBM, no intermediate vector.
 * Returns CWSLG_ERR_MODE for a mode that route rejects ("Unknown mode", :566-570), CWSLG_ERR_NO_FRAME before the
 * first finalised frame.  cwslg_decoder_block_field() gives offset/size of "ipc","ss","savg","sred","d2","params"
 * or of any params member by name (e.g. "nzhsym"). */
size_t cwslg_decoder_block_bytes(int js8);
int cwslg_decoder_block_field(int js8, const char *name, size_t *offset, size_t *bytes);
int cwslg_fill_decoder_block(cwslg_ctx *ctx, int ch_id, void *block, size_t block_bytes, int js8, int decodedepth,
                             int highest_decode_hz, uint64_t *start_epoch);
/* 1 = shared memory, 0 = wave file: the route DecoderPool gives an item of this mode (:379-395; WSPR, JS8 and the
 * FST4 family always go through a file). */
int cwslg_decoder_route(const char *mode, int transfer_shmem);
/* Program name ("jt9.exe" / "wsprd.exe" / "js8.exe") and argument string, character for character as
 * DecoderPool.hpp:634-659 (shared memory; target = key) and :1007-1046 (wave file; target = file name) build them. */
int cwslg_decoder_command(const char *mode, int shmem_route, int numjt9threads, int decodedepth, int highest_decode_hz,
                          int wspr_cycles, float trperiod, const char *target, char *app, size_t app_cap, char *opts,
                          size_t opts_cap);
/* The same frame as 12 kHz float audio BEFORE prepareAudio's scaling (for the 1e-5 check). */
int cwslg_fetch_audio_f32(cwslg_ctx *ctx, int ch_id, float *dst, size_t cap, size_t *n_valid);
/* Device pointers of the last finalised frame (valid until the next boundary of that channel). */
int cwslg_frame_device_ptrs(cwslg_ctx *ctx, int ch_id, const int16_t **d_i16, const float **d_f32);
/* Sync candidates of the last finalised frame (FT8/FT4 channels with sync enabled). */
int cwslg_enable_sync(cwslg_ctx *ctx, int enable, float syncmin, int max_cand, int f_lo_hz, int f_hi_hz);
/* Final ORDER and CUT of the FT8 / FT4 candidate lists (ABI 5).  Upstream's source is not in the reference tree, so which of the two a given
 * jt9 build does cannot be verified here (PARITY UNPINNED); both are implemented -- oracle, kernels and tests/indep_sync.py -- bit-identically:
 *   CWSLG_ORDER_SYNC_DESC (default): strongest first, the first max_cand kept.  Believed to mirror sync8.f90's "Sort by sync" lines (commented
 *       out in WSJT-X 2.x, live in 1.x) and getcandidates4.f90's list by height.
 *   CWSLG_ORDER_FREQ_ASC: ascending frequency, the first max_cand IN THAT ORDER kept -- believed to mirror WSJT-X 2.x sync8.f90 ("Sort by
 *       frequency": indexx on the frequency column, copy while k <= maxcand); entries of one bin stay in their order of discovery (the builder's
 *       tie rule: upstream's indexx is not stable).  FT4: the peaks as getcandidates4 finds them scanning upwards.  (Upstream's "nfqso first"
 *       promotion has no counterpart: the skimmer has no QSO frequency.)
 * The two lists hold the same entries unless the list is cut at max_cand; FT4's hold the same entries always (its scan stops at max_cand before
 * any ordering).  Applies from the next boundary on.
 * sync8's near-duplicate rule on the time axis is evaluated in single precision exactly as
 *       tdiff = fabsf(((float)lag_i - 0.5f) * tstep - ((float)lag_j - 0.5f) * tstep) < 0.04f,   tstep = 480 / 12000.0f,
 * under which 77 of the 124 lag pairs exactly one step apart count as duplicates and 47 do not (0.04 is not a binary fraction; the table is pinned
 * in tests/test_sync_oracle.py); a double-precision evaluation would make none of them duplicates.  Also unverifiable here. */
#define CWSLG_ORDER_SYNC_DESC 0
#define CWSLG_ORDER_FREQ_ASC  1
int cwslg_set_candidate_order(cwslg_ctx *ctx, int order);
/* *start_epoch (may be NULL) = start epoch of the frame the list was computed from: compare it with cwslg_fetch_frame's, or use
 * cwslg_fetch_slot, which returns both under one ticket (ABI 5; ItemToDecode carries epochTime with its audio, DecoderPool.hpp:174-210). */
int cwslg_fetch_candidates(cwslg_ctx *ctx, int ch_id, cwslg_candidate *dst, int max, int *n, uint64_t *start_epoch);
/* FT4 channels run getcandidates4's spectral-peak search instead (freq_hz = interpolated peak - 1.5 tone spacings,
 * sync = normalised peak height, time_step/dt_s = 0); its threshold defaults to upstream's 1.2. */
/* FT4 coherent sync (row a13; PARITY UNPINNED, restated from upstream ft4_decode / ft4_downsample / sync4d): every
 * getcandidates4 candidate of an FT4 channel is band-limited to a 666.7 Hz complex baseband around its frequency and
 * correlated with the four 4x4 Costas blocks over ft4_decode's three start-time segments (coarse then fine grid in start
 * sample and frequency tweak).  One record per candidate and segment that passes (sync >= 1.2, not weaker than segment 1,
 * 10 < f1 < 4990), in candidate order then segment order.  Enabled by default whenever sync is enabled. */
typedef struct {
    float   f0_hz;        /* the getcandidates4 candidate                                            */
    float   f1_hz;        /* f0 + best frequency tweak (Hz)                                          */
    float   dt_s;         /* ibest / 666.67 - 0.5                                                    */
    float   sync;         /* sum of the four block correlation magnitudes / 64                       */
    int32_t ibest;        /* start sample of the first Costas block at 666.7 Hz                      */
    int32_t idf;          /* frequency tweak, Hz                                                     */
    int32_t seg;          /* 1..3: ft4_decode's start-time segment                                   */
    int32_t cand;         /* index into the cwslg_fetch_candidates list                              */
} cwslg_ft4_sync;
int cwslg_enable_ft4_coherent(cwslg_ctx *ctx, int enable);
int cwslg_fetch_ft4_sync(cwslg_ctx *ctx, int ch_id, cwslg_ft4_sync *dst, int max, int *n, uint64_t *start_epoch);
int cwslg_fetch_slot(cwslg_ctx *ctx, int ch_id, int16_t *frame, size_t cap, void *list, size_t list_bytes,
                     cwslg_ft4_sync *ft4, int max_ft4, cwslg_slot_result *out);
int cwslg_set_ft4_syncmin(cwslg_ctx *ctx, float syncmin);

/* Intermediate products of the sync stage for parity tests: what = 0 symbol spectra [372][nbins] float,
 * 1 red, 2 red2 (float[1921], before normalisation), 3 jpeak, 4 jpeak2 (int32[1921]).  *n_items = items available.
 * FT4 channels: 0 = [122][1168] windowed spectra, 1 = savsm/sbase (first 1153 entries), 2 = sbase,
 * 5 = frame spectrum (complex float[36289]), 6 = unit-power baseband of candidate 0 (complex float[4032]). */
int cwslg_sync_debug_fetch(cwslg_ctx *ctx, int ch_id, int what, void *dst, size_t cap_bytes, size_t *n_items, int *row_len);

/* ---- candidate search of the 120 s modes (BASELINE.json configs[4]) ----
 * PARITY UNPINNED like the FT8/FT4 stage: the reference hands WSPR frames to `wsprd -C <cycles> -o 5 -d <wav>` and FST4W-120
 * frames to `jt9 -W -p 120 ... -L 1400 -H 1600 -F 200 <wav>` (DecoderPool.hpp:1019-1033); what runs here is the candidate-finding
 * front end of those programs as restated in oracle/longsync_oracle.c.  When enabled, cwslg_slot_boundary() runs it on every
 * WSPR / FST4W-120 frame it finalises.
 *   WSPR:  wsprd.c main(): FFT-based /32 down-conversion to 375 Hz around 1500 Hz, 359 half-symbol 512-point spectra, smoothed
 *          spectrum / noise percentile / local maxima within +-110 Hz ordered by snr, then per candidate the coarse
 *          (frequency, shift, drift) search on the 162-symbol sync vector.  freq_hz is relative to 1500 Hz audio; shift is in
 *          375 Hz samples from the start of the file (wsprd prints DT = shift/375 - 2 s).
 *   FST4W: get_candidates_fst4.f90: comb-summed power spectrum over [nfa, nfb], 30th-percentile normalisation, CLEAN peak pick
 *          above minsync (at most 100), in order of discovery (strongest first). */
typedef struct {
    float   freq_hz;
    float   snr_db;
    float   drift;
    float   sync;
    int32_t shift;
} cwslg_wspr_candidate;
typedef struct {
    float   freq_hz;
    float   snr;            /* peak of the normalised comb spectrum ("rough estimate of SNR") */
    int32_t bin;            /* index on the baud/2 grid */
    int32_t pad_;
} cwslg_fst4w_candidate;
int cwslg_enable_long_sync(cwslg_ctx *ctx, int enable, int fst4w_nfa_hz, int fst4w_nfb_hz, float fst4w_minsync);
int cwslg_fetch_wspr_candidates(cwslg_ctx *ctx, int ch_id, cwslg_wspr_candidate *dst, int max, int *n, uint64_t *start_epoch);
int cwslg_fetch_fst4w_candidates(cwslg_ctx *ctx, int ch_id, cwslg_fst4w_candidate *dst, int max, int *n, uint64_t *start_epoch);
/* Intermediate results for parity tests.  WSPR channels: what = 0 the 375 Hz baseband (complex float[46080]), 1 the spectra
 * (float[359][512], time-major: wsprd's ps[j][i] is element [i][j]), 2 the normalised smoothed spectrum (float[411]).
 * FST4W channels: 3 the normalised comb spectrum s2 (float, indexed by the baud/2 bin), 4 the band of the long transform
 * (complex float, first bin = nint(ina*df2/df1) - ndh). */
int cwslg_long_sync_debug_fetch(cwslg_ctx *ctx, int ch_id, int what, void *dst, size_t cap_bytes, size_t *n_items);

/* ---- introspection for bench / tests ---- */
typedef struct {
    uint64_t demod_launches;       /* demod kernel launches                                   */
    uint64_t demod_samples;        /* complex input samples consumed, summed over channels    */
    uint64_t finalize_launches;
    uint64_t sync_launches;
    uint64_t frames_emitted;
    uint64_t frames_discarded;     /* startEpochTime == 0                                      */
    uint64_t blocks_dropped;       /* "af buffer full" events (Instance.cpp:268-271)           */
    uint64_t h2d_bytes;
    double   demod_ms;             /* HIP-event time of demod kernels on the context stream   */
    double   finalize_ms;
    double   sync_ms;
    uint64_t phasor_regrows;       /* checkpoint tables extended because a channel kept discarding frames */
    uint64_t rendezvous_calls;     /* slot boundaries that went through the multi-GPU rendezvous          */
    uint64_t rendezvous_frames;    /* frames over ALL processes at the last rendezvous                    */
    uint64_t rccl_world;           /* ranks of the built-in RCCL communicator (cwslg_rccl_init), 0 without one */
    uint64_t rendezvous_flags_and; /* AND over all ranks of cwslg_set_rendezvous_flag's value at the last built-in rendezvous */
    /* ABI 4 */
    double   demod_clock_mhz;      /* shader clock INSIDE the timed demod launches since the last reset: mean over launches of delta s_memtime /
                                    * delta s_memrealtime x 100 MHz, read by one workgroup at the start and the end of its life (exact mode: a
                                    * persistent workgroup, i.e. the whole launch; fast mode: the tile workgroup in the middle of the grid;
                                    * 0 until a timed launch has been drained).  What bench.py prices roofline.valu_pipe at.              */
    uint64_t demod_clock_launches; /* launches that contributed to it                                                                 */
    uint64_t push_calls;           /* host pushes accepted (cwslg_push_iq: one per call; cwslg_push_iq_many: one per receiver)         */
    uint64_t push_batches;         /* cwslg_push_iq_many calls                                                                        */
    double   push_host_ms;         /* wall time spent inside host pushes (staging copy + enqueue), summed over the calling threads      */
    double   sync_spectra_ms;      /* of sync_ms: the FT8 symbol-spectra kernel ...                                                     */
    double   sync_search_ms;       /* ... and the FT8 Costas search + candidate selection                                                */
    /* ABI 5 */
    uint64_t demod_blocks_read;    /* blocks (D input samples = one 12 kHz output each) the demod launches fetched and put through the arithmetic,
                                    * summed over channels: the pending blocks plus every stream's 32-block warm-up (exact) / every tile's
                                    * 31-block history (fast).  x D / demod_samples = redundancy.                                          */
    uint64_t process_deferred;     /* cwslg_process() calls that returned without a launch (cwslg_set_process_threshold)                    */
} cwslg_stats;
int cwslg_get_stats(cwslg_ctx *ctx, cwslg_stats *out);
int cwslg_reset_stats(cwslg_ctx *ctx);
/* Enable HIP-event timing of every kernel launch (bench.py's roofline leg).  Off by default. */
int cwslg_set_timing(cwslg_ctx *ctx, int enable);
/* Name (with template arguments) of the kernel the context's most recent demod launch ran -- what bench.py reports as
 * roofline.kernel; "" before the first launch.  The pointer is to a string literal. */
const char *cwslg_demod_kernel_name(cwslg_ctx *ctx);
/* Raw stream handle (hipStream_t) so callers can order their own work (torch, RCCL) against it. */
void *cwslg_stream(cwslg_ctx *ctx);
/* Host-side DSP constants exactly as uploaded (tests pin them against the oracle):
 * taps[32*D], tone[2*D] (re,im), phase_inc[2]; returns D (= Fs/12000) or <0. */
int cwslg_channel_constants(cwslg_ctx *ctx, int ch_id, float *taps, float *tone_ri, float *phase_inc_ri);
/* Phasor checkpoints as computed on the device: entry c is phase_{c*stride}, stride = cwslg_phasor_checkpoint_stride()
 * blocks (one block = one 12 kHz output sample).  Copies up to n complex values. */
int cwslg_phasor_checkpoint_stride(void);
int cwslg_channel_phasor_checkpoints(cwslg_ctx *ctx, int ch_id, float *dst_ri, size_t n, size_t *n_total);

#ifdef __cplusplus
}
#endif
#endif /* CWSL_GPU_H */

/*
 * sdrx_gpu.h -- C ABI of libsdrxgpu: the MI355X (gfx950) channel bank that replaces the
 * per-channel channelizer loop of johanhedin/sdrx.
 *
 * What it replaces (reference file:line):
 *   - the loop over channels in data_cb, src/sdrx.cpp:327-346, i.e. one MSD::decimate
 *     (src/msd.hpp:75-159) per channel per 32 ms block, or one DS::addJob (src/ds.hpp:43)
 *     / ThreadPool::submit (src/tp.hpp:50) per channel with -t;
 *   - the per-sample IF-AGC -> AM envelope -> LF-AGC lines of alsa_write_cb,
 *     src/sdrx.cpp:455-463 (AGC::adjust src/agc.hpp:48-69, Demod::demod src/sdrx.cpp:178-180,
 *     LfAGC::adjust src/agc.hpp:106-131);
 *   - the per-channel set-up of src/sdrx.cpp:1680-1714 (translator table, MSD, AGC parameters).
 * One sdrxgpu_bank = all channels of one IQ stream on one GPU; one sdrxgpu_bank_process call
 * = one batched pass of the whole bank over one IQ block (any length).  All persistent DSP
 * state (what MSD keeps in its delay lines and trans_pos_, msd.hpp:428-440; AGC gain/power,
 * agc.hpp:71-78) lives inside the bank and carries across calls with the reference's
 * streaming semantics (zero-initialised history, k-th output when the (k+1)*M-th input arrives).
 *
 * Plain C: no C++ or torch types cross this boundary.  IQ samples are interleaved
 * {re, im} float32 = std::complex<float> = reference iqsample_t (src/iqsample.hpp:26).
 *
 * Devices: every entry makes the bank's device current for its own duration and restores the caller's current HIP device on
 * return.
 *
 * Threading: a bank is driven by one thread at a time (data_cb is never re-entered,
 * sdrx.cpp:306); different banks may be driven concurrently from different threads, each
 * on its own HIP stream.  There is no global mutable state outside banks but a count of the
 * banks alive per device (read when a bank is created and once more when its first block
 * is processed, to pick between two kernel sequences that give the same bits).
 *
 * NUMERICAL CONTRACT (one table; INTEGRATION.md 6 has the measurements, tests/test_gpu_pairs_stress.py enforces it on fixtures
 * from the compiled reference).  All paths compute the reference's chain in exact fp32 operations; they differ in the ORDER of
 * summation, as the reference's own two builds do (Release: AVX2 + FMA; portable: source order).  "spread" = the distance
 * between those two builds on the case at hand; comparator |got - ref| / max(|ref|, rms of the channel), ref = portable build.
 *
 *   path                     runs when                                   <= 1e-5 where        never beyond    cut-       layout-
 *                                                                        spread <=            k x spread      invariant  invariant
 *   vector kernels           < 1 024 channels, few +k/-k pairs, every    0.5e-5               1.7 (<= 1.41)   yes        yes
 *    (reference's order)     chain but 10 MS/s, or
 *                            SDRXGPU_FLAG_REFERENCE_ORDER
 *   k_stage12m               10 MS/s, >= 1 024 channels mostly in        0.5e-5 (worst case   2.1 (<= 1.75)   yes        within the
 *    (f32 matrix pipe)       +k/-k pairs: automatic                      1.06e-5 / 1.27e-5)                              family (*)
 *   k_stage123m              only SDRXGPU_FLAG_FUSED_STAGE3              0.3e-5               3.5 (<= 3.15)   no (to     within the
 *    (+ stages 2-3 fused)                                                                                     tolerance) family
 *   k_stage12m6              only tuning.matrix_pipe = 2 on a 6 MS/s     -- (1.38e-5 /        4 (<= 3.43)     yes        within the
 *    (6 MS/s, matrix pipe)   bank                                        1.78e-5 worst)                                  family
 *
 *   cut-invariant: the same bits however the stream is cut into calls.  layout-invariant: a channel's bits do not depend on who
 *   else is in the bank, on the bank's size or on how a channel list is sharded over GPUs.  (*) which family a bank takes depends
 *   on its channel list; pin tuning.matrix_pipe where bits must agree across layouts (a sdrxgpu_multibank pins it by itself).
 *   SDRXGPU_FLAG_REFERENCE_ORDER is THE way to hold 1e-5 wherever the reference's own builds allow it, on every channel, for 15 %
 *   of a 10 000-channel block.  On BASELINE's synthetic inputs every path is <= 4.7e-6.  Audio with SDRXGPU_FLAG_LF_AGC_ACTIVE is
 *   ill-conditioned in the reference itself (its two builds are 1.7e-5 apart after 5 blocks): measured <= 1.4e-5 block by block.
 *
 * Errors: every function returns 0 on success or a negative SDRXGPU_ERR_* code, in the
 * style of the reference's device layer (src/r820_dev.hpp:61-72).  There is NO CPU fallback:
 * without a usable gfx950 device sdrxgpu_bank_create fails with SDRXGPU_ERR_NO_DEVICE.
 */
#ifndef SDRX_GPU_H
#define SDRX_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDRXGPU_OK                 0
#define SDRXGPU_ERR_INVALID_ARG   -1
#define SDRXGPU_ERR_NO_DEVICE     -2
#define SDRXGPU_ERR_NO_MEM        -3
#define SDRXGPU_ERR_HIP           -4
#define SDRXGPU_ERR_UNSUPPORTED   -5   /* stage chain / translator length the kernels do not cover */
#define SDRXGPU_ERR_OUT_TOO_SMALL -6   /* out_stride smaller than the outputs this call produces  */

/* sdrxgpu_bank_cfg.modulation[] values (Channel::mod, sdrx.cpp:151/230) */
#define SDRXGPU_MOD_AM 0
#define SDRXGPU_MOD_FM 1

/* sdrxgpu_bank_cfg.flags */
#define SDRXGPU_FLAG_LF_AGC_ACTIVE 0x1u  /* --lf-agc: LfAGC::activate(), sdrx.cpp:1710 */
#define SDRXGPU_FLAG_FTFIR         0x4u  /* --ftfir (EXPERIMENTAL in the reference, sdrx.cpp:1199): MSD's use_ftfir, msd.hpp:104-131 */
#define SDRXGPU_FLAG_PIPELINED     0x2u  /* throughput mode for sdrxgpu_bank_process_device only: call k launches the
                                            mixer + stages 1-2 of block k concurrently with the remaining stages and
                                            the AGC/demod of block k-1, and returns the outputs of block k-1 (one block
                                            = 32 ms of latency, like one more chunk in sdrx's ring of 8, sdrx.cpp:1742);
                                            sdrxgpu_bank_flush returns the last block.  Same outputs, bit for bit.
                                            The two halves are two chains of work that do not wait for each other inside a
                                            call -- except that the bank's stream waits for the copy of d_iq_in into the bank's
                                            ingest ring: a caller's own input buffer may be rewritten in the bank's stream order
                                            right after the call, as with the other entries.  A block written straight into the
                                            ingest ring (sdrxgpu_bank_ingest_buffer: no copy) is read by this call's kernels until
                                            call k+1 has been enqueued; the bank's OTHER ring is the place for block k+1.       */

#define SDRXGPU_FLAG_FUSED_STAGE3   0x8u  /* large banks of the 10 MS/s chain (where the mixer + stage 1 run on the f32 matrix pipe): run
                                            decimator stages 2 and 3 in that kernel too, as one 119-tap filter, and keep no stage-2
                                            scratch (k_stage123m + k_tail_c): 11 % less time per block, 2.5 GB -> 1.0 GB of HBM traffic.
                                            The price is rounding: a chain of 119 fp32 products rounds relative to partial sums that
                                            carry every signal still in band after stage 1, so on a channel NEXT to a strong carrier
                                            this path is 2 to 3 x the reference's own Release-vs-portable deviation away from the
                                            reference where the default path is 1.1 to 1.4 x (at the edge of 1e-5 under a carrier of 0.3 of
                                            full scale over a floor of 0.25, beyond it under one of 0.6; profiles/r04_b_parity_pairs.jsonl),
                                            and a different cut of the stream into calls changes the last bits.  Off by default. */

#define SDRXGPU_FLAG_REFERENCE_ORDER 0x10u /* only kernels that add every FIR in the reference's own order (msd.hpp:309-331: folded, outside
                                            in, mixer product first) -- the vector kernels, which are bit-identical among themselves
                                            whatever the bank's size, channel list or sharding.  Without it a bank of >= 1 024 channels
                                            whose channels come in +k / -k pairs takes the f32 matrix pipe for the mixer + stage 1
                                            (k_stage12m: 15 % less time per 10 000-channel block), whose P -+ Q formulation is 1.2 to
                                            1.75 x the reference's own Release-vs-portable deviation away from the reference where the
                                            vector kernels are 0.9 to 1.4 x: within 1e-5 wherever the reference's two builds agree to
                                            0.5e-5, and up to 1.1e-5 in two of the 30 stress cases where the vector kernels hold
                                            1e-5 (the weak partner of a carrier at 0.9 of full scale over a floor of 0.05: 1.06e-5
                                            against 6.4e-6).  Same as tuning.matrix_pipe = 1, as a
                                            flag an application sets. */

typedef struct sdrxgpu_bank sdrxgpu_bank;

/* std::complex<float> / iqsample_t */
typedef struct { float re, im; } sdrxgpu_iq;

/* MSD::Stage (src/msd.hpp:46-49): decimation factor + odd-length symmetric low-pass taps. */
typedef struct {
    uint32_t     m;
    uint32_t     n_taps;
    const float *taps;        /* copied at create, like MSD::S does (msd.hpp:182-187) */
} sdrxgpu_stage;

/* Per channel: the translator ("tuner") table MSD's constructor takes (msd.hpp:54), built by
 * the caller with the reference formula (sdrx.cpp:1686-1689; sdrxgpu_make_translator does it).
 * n_translator == 0 <=> channel at the tuner centre, no mixing (msd.hpp:79-101).  All non-empty
 * tables of one bank must have the same length N (N is per sample rate, sdrx.cpp:1565-1587). */
typedef struct {
    const sdrxgpu_iq *translator;   /* copied at create */
    uint32_t          n_translator;
} sdrxgpu_channel;

/* AGC / LfAGC parameters (src/agc.hpp:30-35, 85-89) plus the initial state the constructor
 * leaves behind.  sdrxgpu_agc_params_sdrx() fills the values sdrx uses. */
typedef struct {
    float attack, decay, reference, max_gain, alpha;
    float init_gain, init_power;
} sdrxgpu_agc_params;

/* Geometry / kernel-path overrides for tests and tuning runs.  NULL in sdrxgpu_bank_cfg.tuning (the default) =
 * everything automatic.  Every choice but matrix_pipe computes the same outputs bit for bit
 * (tests/test_gpu_parity.py); a choice whose tiles do not fit the CU's LDS, or that is not compiled for the chain,
 * makes sdrxgpu_bank_create fail. */
typedef struct {
    uint32_t struct_size;    /* = sizeof(sdrxgpu_tuning): checked by sdrxgpu_bank_create (fields have come and gone
                                between versions of this header; a caller built against another layout is refused
                                instead of having its values read as other fields)                            */
    uint32_t k12_waves;      /* waves per workgroup of the stage-1/2 kernel: 4, 8 or 16; 0 = auto            */
    uint32_t k12_cpw;        /* channels per wave of that kernel: 1, 2 or 4; 0 = auto                         */
    uint32_t tail_tile;      /* final outputs per workgroup of the separate tail kernel, 8..128; 0 = auto     */
    uint32_t tail_generic;   /* 1 = run-time-tap tail kernel even where a compile-time one exists             */
    uint32_t fused_tail;     /* stage 3 inside the VECTOR stage-1/2 kernel (k_stage12 / k_stage12p + k_tail4): 0 = auto
                                (banks of 6 000 channels and more that stay on the vector kernels -- e.g. a channel list
                                without conjugate partners; a bank that takes the matrix pipe runs k_stage12m + k_tail_s
                                whatever its size), 1 = never, 2 = require it (and with it the vector kernels)        */
    uint32_t host_parts;     /* channel parts of sdrxgpu_bank_process (D2H of one beside the kernels of the next):
                                0 = auto, 1..8                                                                */
    uint32_t pair_lanes;     /* two channels per lane of the stage-1/2 kernel (conjugate partners share the mixer):
                                0 = auto, 1 = never, 2 = require                                              */
    uint32_t matrix_pipe;    /* mixer + stage 1 of the 10 MS/s chain on the f32 matrix pipe (k_stage12m: exact fp32
                                fma chains in another summation order -- the one choice here whose outputs differ
                                from the others', by rounding only): 0 = auto (banks of 1 024 channels or more whose
                                channels mostly come in +k / -k pairs: the kernel's work goes with pairs + unpaired
                                channels, and bank_create compares the two kernels' launch counts; with
                                SDRXGPU_FLAG_FUSED_STAGE3: k_stage123m), 1 = never, 2 = require k_stage12m -- or, for
                                the 6 MS/s chain (15,53)(5,21), k_stage12m6: never automatic, 0.93 -> 0.58 ms per
                                10 000-channel block for up to 3.4 x the reference's own rounding spread --,
                                3 = require k_stage123m (stages 2-3 in the same kernel).  Because the automatic
                                choice depends on the bank's channel list, a channel's last bits depend on it too -- and
                                so on how a channel list is sharded over GPUs (<= the tolerances of INTEGRATION.md 6): pin
                                this field where bit-identical results across layouts matter                     */
    uint32_t m_slot_groups;  /* k_stage12m workgroup = m_slot_groups x m_seg_groups tiles (waves) of 16 channel    */
    uint32_t m_seg_groups;   /*   pairs x 16 time segments: (2,2), (4,2) or (2,4); 0 = auto                        */
    uint32_t tail_agc;       /* the remaining decimator stage(s) and the AGC / demodulator as ONE kernel (k_tail_agc: tiles of 32
                                outputs handed from producer waves to the AGC waves through LDS; same bits as k_tail_s + k_agc):
                                0 = auto (banks of 1 .. 4 channels per CU -- 256 .. 1 024 on an MI355X -- on the separate tail that are
                                the only bank on their device when their FIRST BLOCK is processed: 1 to 7 % less time per block, sdrxgpu.hip
                                SDRX_TAILAGC_AUTO_*; below, the two kernels' own latency chains are shorter than its pipeline's
                                fill, above, its workgroups no longer fit the chip at once; beside another bank its long-lived
                                workgroups are in that bank's way: DESIGN.md 4.10), 1 = never, 2 = require.  The choice is made
                                once -- provisionally at create, finally with the bank's first block, when an application has
                                created all its banks -- and kept: sdrxgpu_bank_info.tail_agc says which sequence runs        */
    uint32_t tail_agc_cpg;   /*   channels per workgroup of that kernel: 1, 2 or 4; 0 = auto                           */
    uint32_t y2_keep_mb;     /* k_stage12m's scratch stores: the sets of a launch written last -- this many MB + 1 of stage-2
                                outputs -- are ordinary stores (the tail kernel reads them first, out of the Infinity Cache),
                                everything before them goes out with the streaming hint (same bits):
                                0 = auto (192 MB), 1 = everything streams, 0xFFFFFFFF = nothing does                   */
    uint32_t host_part_ch[7];/* explicit channel boundaries of the host parts, both cuts (upper end of part 0, 1, ...; rounded down to
                                whole channel groups; at most host_parts parts; the last part ends with the bank): 0 = automatic */
} sdrxgpu_tuning;

typedef struct {
    uint32_t               struct_size;  /* = sizeof(sdrxgpu_bank_cfg) */
    uint32_t               flags;        /* SDRXGPU_FLAG_* */
    uint32_t               n_stages;     /* 3 or 4 (every chain of sdrx.cpp:1588-1672) */
    const sdrxgpu_stage   *stages;
    uint32_t               n_channels;
    const sdrxgpu_channel *channels;
    sdrxgpu_agc_params     agc;          /* IF AGC on the decimated IQ          */
    sdrxgpu_agc_params     lf_agc;       /* LF AGC on the demodulated envelope  */
    int32_t                device;       /* HIP device ordinal, -1 = current    */
    void                  *stream;       /* hipStream_t to run on; NULL = the bank creates its own */
    uint32_t               max_in_len;   /* largest in_len of one call the buffers are sized for;
                                            0 = 32 ms block of this chain's rate * 1 (longer calls
                                            are split internally)                */
    uint32_t               block_hint;   /* 0 = auto; else bodies per lane of the stage-1/2 kernel
                                            (must divide the bodies per segment; 0xFFFFFFFF = a whole segment) */
    /* Optional audio band-pass on every channel's demodulated envelope: FIR::filter with setGain
     * (src/fir.hpp:79-127; sdrx runs FIR2 with coeff_bp4am_channel, coeffs.hpp:1812, on the mixed stereo
     * sum, sdrx.cpp:646 -- linear, so per-channel mono filtering before the mix is the same thing).
     * n_audio_fir_taps == 0: off.  Taps are copied at create; gain in dB as FIR::setGain takes it. */
    const float           *audio_fir_taps;
    uint32_t               n_audio_fir_taps;
    float                  audio_fir_gain_db;
    /* Optional squelch, reference src/sdrx.cpp:533-573 + the gating of the audio by it, sdrx.cpp:458-503.
     * sql_level == NULL: off (audio as if the squelch were open, the default of this library).  Otherwise one
     * level in dB per channel (Channel::sql_level, 0 = always open); every sdrxgpu_bank_process* call must then
     * produce exactly 512 outputs per channel (FFT_SIZE = CH_IQ_BUF_SIZE, sdrx.cpp:129-130).  Per chunk: the
     * audio is gated by the state the PREVIOUS chunk left (ramp_up on the chunk after opening, ramp_down on the
     * chunk after closing, silence and a frozen LF-AGC while closed), then the state is updated from this
     * chunk's pre-AGC IQ.  Tables are the caller's (coeffs.hpp:2018/2541/3064), 512 floats each, copied. */
    const float           *sql_level;
    const float           *passband_shape;
    const float           *ramp_up;
    const float           *ramp_down;
    /* Optional stereo mix-down of all channels' (gated) audio, the back half of alsa_write_cb: pan weights by
     * Channel::pos (sdrx.cpp:469-497), the common FIR2 band-pass with its volume (sdrx.cpp:646, 880-881;
     * fir.hpp:131-183) and the float -> S16 conversion (sdrx.cpp:649-656).  pan_pos == NULL: off.  pan_pos[c] in
     * -2..2 as get_audio_pos returns it (sdrx.cpp:1024-1044; sdrxgpu_audio_pos); n_mix_fir_taps == 0: no filter.
     * Each sdrxgpu_bank_process* call (at most max_in_len samples) then also leaves one stereo buffer on the
     * device, read with sdrxgpu_bank_get_mix.  Channels are summed in channel order inside groups of 64 and the
     * groups in order: identical to the reference's order for banks of <= 64 channels. */
    const int8_t          *pan_pos;
    const float           *mix_fir_taps;
    uint32_t               n_mix_fir_taps;
    float                  mix_fir_gain_db;
    /* Optional per-channel modulation (Channel::mod, sdrx.cpp:230; --modulation / "ch/sql/FM", EXPERIMENTAL in the
     * reference): NULL = every channel AM.  SDRXGPU_MOD_FM channels get Demod::demod's FM branch (sdrx.cpp:181-194)
     * in place of std::abs: audio = atan2 of the phase step between consecutive amplitude-normalised AGC outputs. */
    const uint8_t         *modulation;
    const sdrxgpu_tuning  *tuning;       /* NULL = automatic (what a host application passes) */
} sdrxgpu_bank_cfg;

typedef struct {
    uint32_t n_channels;
    uint32_t m_total;          /* MSD::m(), msd.hpp:71 */
    uint32_t n_translator;     /* N, 0 if no channel mixes */
    uint32_t history_len;      /* input samples of history kept on the device */
    uint64_t samples_in;       /* IQ samples consumed since create/reset */
    uint64_t samples_out;      /* outputs produced per channel since create/reset */
    uint64_t device_bytes;     /* HBM held by the bank */
    uint32_t fused_tail;       /* 1: stage 3 runs inside the stage-1/2 kernel (large banks of the 4-stage chains) */
    uint32_t pair_lanes;       /* > 0: two channels per lane in the stage-1/2 kernel; the number of conjugate pairs + 1 */
    uint32_t k12_waves;        /* geometry of the stage-1/2 launch as chosen (or forced by tuning): waves per workgroup, */
    uint32_t k12_cpw;          /*   channels (slots) per wave, */
    uint32_t k12_bodies;       /*   bodies per lane */
    uint32_t matrix_pipe;      /* 1: mixer + stage 1 run on the f32 matrix pipe (k_stage12m); 2: and stages 2-3 in the same kernel (k_stage123m) */
    uint32_t m_slots;          /*   its channel-pair slots (pairs + channels without partner + padding) */
    uint32_t tail_agc;         /* > 0: tail stages + AGC/demod run as one kernel (k_tail_agc); the channels per workgroup */
} sdrxgpu_bank_info;

/* Fill the AGC / LfAGC parameter blocks as sdrx configures them per channel
 * (default-constructed objects + the setters of src/sdrx.cpp:1702-1710). */
void sdrxgpu_agc_params_sdrx(sdrxgpu_agc_params *agc, sdrxgpu_agc_params *lf_agc);

/* Host-side translator table, verbatim the reference formula src/sdrx.cpp:1686-1689:
 * t[n] = exp(complex<float>(0, float(-2*pi*n*ch_offset*z/N))), n = 0..N-1. */
int sdrxgpu_make_translator(int N, int z, int ch_offset, sdrxgpu_iq *out);

/* ---- channel names -> translator offsets (host-side set-up; no GPU involved) -------------------------------
 * Aeronautical channel name "NNN.NNN" (8.33 kHz and 25 kHz schemes share one grid: 12 steps per 100 kHz, the last
 * two digits name the step, src/sdrx.cpp:1456-1466) -> frequency in Hz, as parse_fq(str, true) of
 * src/sdrx.cpp:957-1021.  SDRXGPU_ERR_INVALID_ARG for anything parse_fq answers with 0. */
int sdrxgpu_parse_channel(const char *name, uint32_t *fq_hz);
/* Tuner centre frequency sdrx picks for a set of channels: midpoint of the lowest and the highest channel
 * (by name order, Channel::operator<, src/sdrx.cpp:212), rounded to the nearest 100 kHz (src/sdrx.cpp:1398-1408). */
int sdrxgpu_tuner_fq(const char *const *names, size_t n, uint32_t *tuner_fq);
/* channel_to_offset of src/sdrx.cpp:1449-1478: signed number of 8.33 kHz steps of the channel from the tuner
 * centre -- the ch_offset that sdrxgpu_make_translator takes. */
int sdrxgpu_channel_to_offset(const char *name, int32_t tuner_fq, int *offset);
/* verify_requested_bandwidth of src/sdrx.cpp:1424-1443: *fits = 1 iff highest - lowest channel <= 80 % of the sample
 * rate (what sdrx refuses to start without --bw-override, src/sdrx.cpp:1560-1564). */
int sdrxgpu_channels_fit_bandwidth(const char *const *names, size_t n, uint32_t sample_rate_hz, int *fits);

/* Replaces the per-channel set-up of src/sdrx.cpp:1680-1714 (MSD ctor msd.hpp:54-68). */
int sdrxgpu_bank_create(const sdrxgpu_bank_cfg *cfg, sdrxgpu_bank **out_bank);

/* Replaces teardown (sdrx.cpp:1816-1822). */
int sdrxgpu_bank_destroy(sdrxgpu_bank *bank);

/* Back to the freshly-constructed state (zero history, trans_pos 0, initial AGC state). */
int sdrxgpu_bank_reset(sdrxgpu_bank *bank);

/* One block through every channel: replaces the loop of src/sdrx.cpp:327-346 plus lines 455-463.
 *   iq_in     : in_len samples, HOST memory, borrowed for the duration of the call only
 *               (the device layer flips its double buffer right after, airspy_dev.cpp:410-414);
 *   iq_out    : HOST, [n_channels][out_stride]; channel c's decimated IQ starts at
 *               iq_out + c*out_stride exactly as data_cb lays out a ring chunk
 *               (iq_buf_ptr += CH_IQ_BUF_SIZE, sdrx.cpp:338/344); may be NULL;
 *   audio_out : HOST, [n_channels][out_stride] float: LfAGC(|AGC(iq)|), the demodulated
 *               16 kS/s envelope (squelch taken as open); may be NULL;
 *   out_len   : outputs produced per channel by this call (MSD's out_len_ptr, msd.hpp:158).
 * Outputs are complete in caller memory when the call returns. */
int sdrxgpu_bank_process(sdrxgpu_bank *bank, const sdrxgpu_iq *iq_in, size_t in_len,
                         sdrxgpu_iq *iq_out, float *audio_out, size_t out_stride,
                         size_t *out_len);

/* Same with DEVICE pointers; asynchronous on the bank's stream (use sdrxgpu_bank_synchronize
 * or stream order).  d_iq_in may be the pointer returned by sdrxgpu_bank_ingest_buffer, in
 * which case no copy is made (H2D / RCCL broadcast lands directly in the bank's buffer). */
int sdrxgpu_bank_process_device(sdrxgpu_bank *bank, const void *d_iq_in, size_t in_len,
                                void *d_iq_out, void *d_audio_out, size_t out_stride,
                                size_t *out_len);

/* The throughput entry between the two: iq_in is PAGE-LOCKED host memory (hipHostMalloc / sdrxgpu_host_register),
 * the outputs are device pointers, and nothing waits: the H2D copy is queued on the bank's copy stream into the
 * input ring the previous block's kernels are not reading (so it runs beside them), this block's kernels follow on
 * the bank's stream.  iq_in must stay unchanged until the NEXT call on this bank has begun (it waits for the copy
 * if it has not finished -- the device layer's double buffer gives exactly that, airspy_dev.cpp:410-414) or
 * sdrxgpu_bank_synchronize has returned.  Several banks (config 5: 16 streams) driven this way from one or more
 * threads overlap each other's copies and kernels. */
int sdrxgpu_bank_process_pinned(sdrxgpu_bank *bank, const sdrxgpu_iq *iq_in, size_t in_len,
                                void *d_iq_out, void *d_audio_out, size_t out_stride, size_t *out_len);

/* Pipelined banks: outputs of the block whose tail is still pending (device pointers, bank's stream). */
int sdrxgpu_bank_flush(sdrxgpu_bank *bank, void *d_iq_out, void *d_audio_out, size_t out_stride, size_t *out_len);

/* Device address where the NEXT block may be written directly, and its capacity in samples.  The bank has two
 * input rings and uses them alternately (each call of at most max_in_len samples flips), so that block k+1 can
 * arrive -- H2D copy, RCCL broadcast -- while the kernels of block k still read the other ring. */
int sdrxgpu_bank_ingest_buffer(sdrxgpu_bank *bank, void **d_ptr, size_t *capacity);
/* The same for the block AFTER a next call of in_len samples (in_len <= max_in_len): where block k+1 may be
 * written before, or while, block k is processed.  Passing these addresses to sdrxgpu_bank_process_device makes
 * the call copy-free; any other address is copied in, as before. */
int sdrxgpu_bank_ingest_buffer_after(sdrxgpu_bank *bank, size_t in_len, void **d_ptr);

/* Page-lock caller memory that is handed to sdrxgpu_bank_process again and again (the IQ double buffer of the
 * device layer, the chunks of the CRB ring, reference src/crb.hpp:56): H2D/D2H then go by DMA straight from/to
 * it instead of through the runtime's staging buffer.  Thin wrappers so that the caller needs no HIP headers. */
int sdrxgpu_host_register(void *ptr, size_t bytes);
int sdrxgpu_host_unregister(void *ptr);

int sdrxgpu_bank_synchronize(sdrxgpu_bank *bank);
int sdrxgpu_bank_get_info(sdrxgpu_bank *bank, sdrxgpu_bank_info *info);
/* The channel parts sdrxgpu_bank_process (host pointers) runs the bank in, so that the D2H copy of one part's rows crosses PCIe
 * while the kernels of the next run: bounds[0] = 0 < bounds[1] < ... < bounds[*n_parts] = n_channels (cap >= *n_parts + 1 entries,
 * at most 9).  A bank has two cuts: audio_only = 0, for calls that take the decimated IQ (with or without the audio), and
 * audio_only = 1, for calls with iq_out == NULL (a third of the bytes to copy: fewer, larger parts).  One part = the whole bank
 * in one go (banks under 2 048 channels; banks with squelch, audio FIR or mix-down). */
int sdrxgpu_bank_get_host_parts(sdrxgpu_bank *bank, int audio_only, uint32_t *bounds, size_t cap, size_t *n_parts);

/* ---- one channel list over several GPUs of one node, in ONE process --------------------------------------------
 * sdrx is one process with one device thread calling data_cb (src/sdrx.cpp:306-364); its only parallelism is one
 * independent worker per channel (DS, src/ds.hpp:43-77).  A sdrxgpu_multibank keeps that shape across GPUs: the
 * channel list is cut into contiguous ranges, one per device (sdrxgpu_shard_range's rule), every range is an
 * sdrxgpu_bank of its own on its device with its own state, and ONE call processes one block on all of them:
 *   - the block goes to every device by a host-to-device copy of its own from the caller's buffer (2.56 MB per
 *     32 ms block of 10 MS/s; page-lock the buffer with sdrxgpu_host_register and it is one DMA per device) --
 *     no collective, no traffic between GPUs, no RCCL;
 *   - every device's output rows are copied straight into its slice of the caller's [n_channels][out_stride]
 *     arrays (the chunk data_cb fills, iq_buf_ptr += CH_IQ_BUF_SIZE, sdrx.cpp:338/344);
 *   - one worker thread per device drives its bank (DS's thread-per-worker model with devices as the workers), so
 *     the devices run side by side; the call returns when all are done.
 * Which stage-1/2 kernel family runs (tuning.matrix_pipe: the one choice whose outputs differ by rounding) is
 * decided ONCE, for the whole channel list, and pinned in every shard: a channel's bits do not depend on how many
 * devices the list is cut over, nor on which shard it lands in (tests/test_gpu_multibank.py).
 * Not available through a multibank: the stereo mix-down (pan_pos: a sum over all channels), a caller's stream
 * (cfg.stream belongs to one device), SDRXGPU_FLAG_PIPELINED.  cfg.device is ignored. */
typedef struct sdrxgpu_multibank sdrxgpu_multibank;

typedef struct {
    uint32_t n_shards;         /* devices that hold a (non-empty) channel range */
    uint32_t n_channels;
    uint32_t m_total;
    uint32_t matrix_pipe;      /* stage-1/2 kernel family of EVERY shard, as sdrxgpu_bank_info.matrix_pipe */
    uint64_t samples_in, samples_out;
    uint64_t device_bytes;     /* HBM held, all devices */
} sdrxgpu_multibank_info;

/* Contiguous [*lo, *hi) of shard `index` of `n_shards` over `n_items` items: the first n_items % n_shards shards take
 * one item more.  Pure arithmetic (no device). */
int sdrxgpu_shard_range(uint32_t n_items, uint32_t index, uint32_t n_shards, uint32_t *lo, uint32_t *hi);

/* devices: HIP device ordinals, one shard each, in channel order; the same ordinal may appear more than once (two
 * shards on one GPU: how the tests rehearse the path on a one-GPU box).  n_devices == 0 (devices may be NULL): every
 * visible device.  With more devices than channels the last devices stay unused. */
int sdrxgpu_multibank_create(const sdrxgpu_bank_cfg *cfg, const int32_t *devices, uint32_t n_devices, sdrxgpu_multibank **out);
int sdrxgpu_multibank_destroy(sdrxgpu_multibank *mb);
int sdrxgpu_multibank_reset(sdrxgpu_multibank *mb);

/* sdrxgpu_bank_process over all devices: HOST pointers, outputs complete in caller memory on return.  A refused call
 * (out_stride too small, ...) is refused before any device has been touched.  A call that fails ON a device (a HIP
 * error: the code and text name the shard) can leave the shards at different places of the stream:
 * sdrxgpu_multibank_reset (or destroy) before going on.  Like a bank, a multibank is driven by ONE thread at a time
 * (data_cb's, sdrx.cpp:306): its calls are not reentrant; different multibanks are independent. */
int sdrxgpu_multibank_process(sdrxgpu_multibank *mb, const sdrxgpu_iq *iq_in, size_t in_len, sdrxgpu_iq *iq_out,
                              float *audio_out, size_t out_stride, size_t *out_len);
/* sdrxgpu_bank_process_pinned over all devices: page-locked host block in, per-shard DEVICE output pointers
 * (d_iq_out[s] / d_audio_out[s]: [channels of shard s][out_stride], on shard s's device; either array may be NULL),
 * asynchronous on every shard's stream -- sdrxgpu_multibank_synchronize waits for all of them. */
int sdrxgpu_multibank_process_pinned(sdrxgpu_multibank *mb, const sdrxgpu_iq *iq_in, size_t in_len, void *const *d_iq_out,
                                     void *const *d_audio_out, size_t out_stride, size_t *out_len);
int sdrxgpu_multibank_synchronize(sdrxgpu_multibank *mb);
int sdrxgpu_multibank_get_info(sdrxgpu_multibank *mb, sdrxgpu_multibank_info *info);
/* Shard `index` (0 .. n_shards-1): its bank (owned by the multibank; for the getters -- info, gains, squelch, profile --
 * not for process calls), its device and its channel range.  Any pointer may be NULL. */
int sdrxgpu_multibank_shard(sdrxgpu_multibank *mb, uint32_t index, sdrxgpu_bank **bank, int32_t *device, uint32_t *ch_lo, uint32_t *ch_hi);
/* sdrxgpu_bank_get_agc_state / _get_gains / _get_squelch over all shards, in channel order (what sdrx prints per channel,
 * sdrx.cpp:593-634). */
int sdrxgpu_multibank_get_agc_state(sdrxgpu_multibank *mb, float *agc_gain, float *agc_power, float *lf_gain, float *lf_power);
int sdrxgpu_multibank_get_gains(sdrxgpu_multibank *mb, float *agc_gain, float *lf_agc_gain);
int sdrxgpu_multibank_get_squelch(sdrxgpu_multibank *mb, float *snr_db, unsigned char *state);

/* Squelch SNR (dB) and state (0 closed / 1 open) per channel after the last chunk (what sdrx prints,
 * sdrx.cpp:593-634); either may be NULL. */
int sdrxgpu_bank_get_squelch(sdrxgpu_bank *bank, float *snr_db, unsigned char *state);

/* get_audio_pos of src/sdrx.cpp:1024-1044: stereo position -2..2 of channel channel_no out of num_channels. */
int sdrxgpu_audio_pos(unsigned channel_no, unsigned num_channels);

/* Stereo mix of the last process call (banks created with pan_pos): interleaved frames, slot 0 = the
 * audio_buffer_float[i*2] of sdrx.cpp:472, as float (what FIR2 returned) and/or as S16 (what snd_pcm_writei
 * takes, sdrx.cpp:659).  Either pointer may be NULL; *n_frames = frames available (512 per 32 ms block). */
int sdrxgpu_bank_get_mix(sdrxgpu_bank *bank, float *mix_f32, int16_t *mix_s16, size_t cap_frames, size_t *n_frames);
/* The same buffers on the device (valid in stream order after the process call). */
int sdrxgpu_bank_mix_buffers(sdrxgpu_bank *bank, void **d_mix_f32, void **d_mix_s16, size_t *n_frames);

/* AGC gains per channel (what sdrx prints with -v, sdrx.cpp:593-634); either may be NULL. */
int sdrxgpu_bank_get_gains(sdrxgpu_bank *bank, float *agc_gain, float *lf_agc_gain);

/* The complete AGC state per channel: gain_ and power_ of AGC and of LfAGC (agc.hpp:71-78, 133-141) -- the only
 * recurrence state of the path; with samples_in of sdrxgpu_bank_get_info it says where a channel stands.  Any
 * pointer may be NULL. */
int sdrxgpu_bank_get_agc_state(sdrxgpu_bank *bank, float *agc_gain, float *agc_power, float *lf_gain, float *lf_power);

/* Kernel timing with HIP events recorded on the bank's own stream.
 * kernel: 0 = mixer + stages 1-2, 1 = remaining stages, 2 = AGC/demod, 3 = whole process call. */
#define SDRXGPU_K_STAGE12 0
#define SDRXGPU_K_TAIL    1
#define SDRXGPU_K_AGC     2
#define SDRXGPU_K_TOTAL   3
/* enable: 0 off; 1 every kernel and the whole call (8 event records per block, ~20 us of stream time at 10 000
 * channels); 2 only SDRXGPU_K_STAGE12, the dominant kernel (what bench.py keeps on inside its timed region). */
int sdrxgpu_bank_profile_enable(sdrxgpu_bank *bank, int enable);
int sdrxgpu_bank_profile_read(sdrxgpu_bank *bank, int kernel, double *total_ms, uint64_t *launches);
int sdrxgpu_bank_profile_reset(sdrxgpu_bank *bank);

const char *sdrxgpu_strerror(int code);
const char *sdrxgpu_last_error(void);   /* thread-local detail of the last failure */
const char *sdrxgpu_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SDRX_GPU_H */

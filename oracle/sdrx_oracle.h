/*
 * sdrx_oracle.h -- CPU restatement of the sdrx channelizer hot path (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle for the MI355X implementation in sdrx_amd/.  It is
 * NOT part of the product: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product path (libsdrxgpu.so) never links,
 * loads or falls back to anything in oracle/.
 *
 * Every function restates, in plain C99 with the same operation order as the
 * reference's portable (non-AVX2) code path, one piece of johanhedin/sdrx; the
 * reference file:line it follows is cited next to each function in
 * sdrx_oracle.c.  Pinning: oracle/gen_fixtures.py compiles the reference's own
 * header-only classes (oracle/ref_driver.cpp, built into oracle/_ref/) and
 * checks this restatement against them bit-for-bit (portable build) before it
 * writes the golden vectors under tests/golden/.
 *
 * Data layout everywhere: IQ = interleaved {re, im} float32 (reference
 * src/iqsample.hpp:26, std::complex<float>).
 */
#ifndef SDRX_ORACLE_H
#define SDRX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- translator table (reference src/sdrx.cpp:1682-1690) ---- */
void sdrx_oracle_make_translator(int N, int z, int ch_offset, float *out_iq /* 2*N floats */);

/* ---- MSD: multi-stage translating down sampler (reference src/msd.hpp) ---- */
typedef struct sdrx_oracle_msd sdrx_oracle_msd;

sdrx_oracle_msd *sdrx_oracle_msd_create(const float *translator_iq, size_t n_translator,
                                        int n_stages, const uint32_t *m, const uint32_t *n_taps,
                                        const float *const *taps, int use_ftfir);
sdrx_oracle_msd *sdrx_oracle_msd_clone(const sdrx_oracle_msd *src);
size_t sdrx_oracle_msd_m(const sdrx_oracle_msd *msd);
void sdrx_oracle_msd_decimate(sdrx_oracle_msd *msd, const float *in_iq, size_t in_len,
                              float *out_iq, size_t *out_len);
void sdrx_oracle_msd_destroy(sdrx_oracle_msd *msd);

/* ---- AGC / LfAGC (reference src/agc.hpp) ---- */
typedef struct {
    float attack, decay, reference, min_gain, max_gain, alpha, gain, power;
    int   active; /* LfAGC only */
} sdrx_oracle_agc;

void  sdrx_oracle_agc_init(sdrx_oracle_agc *a, float attack, float decay, float reference,
                           float max_gain, float alpha);
void  sdrx_oracle_lfagc_init(sdrx_oracle_agc *a, float attack, float decay, float reference,
                             float max_gain, float alpha);
void  sdrx_oracle_agc_set_max_gain(sdrx_oracle_agc *a, float max_gain);
/* sdrx's per-channel set-up, reference src/sdrx.cpp:1702-1710 */
void  sdrx_oracle_agc_init_sdrx(sdrx_oracle_agc *agc, sdrx_oracle_agc *agc_lf, int use_lf_agc);
void  sdrx_oracle_agc_adjust(sdrx_oracle_agc *a, const float *in_iq, size_t n, float *out_iq);
void  sdrx_oracle_lfagc_adjust(sdrx_oracle_agc *a, const float *in, size_t n, float *out);

/* ---- AM demodulator (reference src/sdrx.cpp:178-180) ---- */
void  sdrx_oracle_demod_am(const float *in_iq, size_t n, float *out);
/* ---- FM demodulator (EXPERIMENTAL in the reference, src/sdrx.cpp:181-194): phase step between consecutive
 *      amplitude-normalised samples; prev[2] = Demod::prev_sample, (0, 0) after construction (sdrx.cpp:175).
 *      Restated from text (Demod lives inside the unbuildable sdrx.cpp): PARITY UNPINNED. ---- */
void  sdrx_oracle_demod_fm(float *prev, const float *in_iq, size_t n, float *out);

/* ---- FIR (reference src/fir.hpp:79-127) ---- */
typedef struct sdrx_oracle_fir sdrx_oracle_fir;
sdrx_oracle_fir *sdrx_oracle_fir_create(const float *c, unsigned n);
void  sdrx_oracle_fir_set_gain(sdrx_oracle_fir *f, float gain_db);
void  sdrx_oracle_fir_filter(sdrx_oracle_fir *f, const float *in, unsigned in_len, float *out);
void  sdrx_oracle_fir_destroy(sdrx_oracle_fir *f);

/* ---- channel bank: the data_cb loop + the per-sample part of alsa_write_cb
 *      (reference src/sdrx.cpp:327-346 and 455-463) over n channels ---- */
typedef struct sdrx_oracle_bank sdrx_oracle_bank;
sdrx_oracle_bank *sdrx_oracle_bank_create(int n_channels, const float *const *translators_iq,
                                          const size_t *n_translator, int n_stages,
                                          const uint32_t *m, const uint32_t *n_taps,
                                          const float *const *taps, int use_ftfir, int use_lf_agc);
/* iq_out[ch*out_stride + i], audio_out[ch*out_stride + i]; either may be NULL.
 * threads <= 1: serial loop over channels (reference default, sdrx.cpp:342-345);
 * threads  > 1: channels split over that many OpenMP threads (models -t / ds.hpp). */
void sdrx_oracle_bank_process(sdrx_oracle_bank *b, const float *in_iq, size_t in_len,
                              float *iq_out, float *audio_out, size_t out_stride,
                              size_t *out_len, int threads);
void sdrx_oracle_bank_destroy(sdrx_oracle_bank *b);

/* ---- squelch + squelch-gated audio of one channel for one 512-sample chunk
 *      (reference src/sdrx.cpp:455-503 audio part, 533-573 squelch part, window 896-900).
 *      PARITY UNPINNED for the FFT: the reference uses FFTW3f (absent here); this restatement evaluates the
 *      same 512-point forward DFT directly in double precision.  Everything around the transform follows the
 *      reference's float arithmetic. ---- */
#define SDRX_ORACLE_FFT_SIZE 512
typedef struct {
    sdrx_oracle_agc agc, agc_lf;
    float sql_level;
    int   sql_state, sql_state_prev;     /* 0 = SQL_CLOSED, 1 = SQL_OPEN (sdrx.cpp:149) */
    float snr;                           /* last SNR in dB */
    int   fm;                            /* Channel::mod: 0 = AM (default), 1 = FM (sdrx.cpp:230) */
    float fm_prev[2];                    /* Demod::prev_sample */
} sdrx_oracle_channel;

void  sdrx_oracle_hamming_window(float *w /* 512 */);
void  sdrx_oracle_channel_init(sdrx_oracle_channel *c, float sql_level, int use_lf_agc);
float sdrx_oracle_squelch_snr(const float *iq /* 512 complex */, const float *window, const float *passband_shape);
/* One chunk: audio[512] (mono, before the pan-mix) with the squelch state left by the previous chunk, then the
 * squelch update from this chunk's (pre-AGC) IQ. */
void  sdrx_oracle_channel_chunk(sdrx_oracle_channel *c, const float *iq, const float *window,
                                const float *passband_shape, const float *ramp_up, const float *ramp_down,
                                float *audio);

/* ---- stereo mix-down of the gated channel audio, the back half of alsa_write_cb ----
 * FIR2 (reference src/fir.hpp:131-183): interleaved stereo FIR; pinned against the reference class through
 * oracle/_ref (tests/test_oracle.py).  audio_pos / pan_mix / to_s16 restate three short blocks of sdrx.cpp that
 * cannot be compiled on their own (static functions inside the ALSA program): PARITY UNPINNED beyond review. */
typedef struct sdrx_oracle_fir2 sdrx_oracle_fir2;
sdrx_oracle_fir2 *sdrx_oracle_fir2_create(const float *c, unsigned n);
void  sdrx_oracle_fir2_set_gain(sdrx_oracle_fir2 *f, float gain_db);
/* in_len counts floats (2 per frame), as the reference's; in and out may be the same array */
void  sdrx_oracle_fir2_filter(sdrx_oracle_fir2 *f, const float *in, unsigned in_len, float *out);
void  sdrx_oracle_fir2_destroy(sdrx_oracle_fir2 *f);
/* get_audio_pos, reference src/sdrx.cpp:1024-1044 */
int   sdrx_oracle_audio_pos(unsigned channel_no, unsigned num_channels);
/* the `switch (ch.pos)` accumulation of sdrx.cpp:469-497 over all channels in channel order:
 * mix[i*2], mix[i*2+1] += w(pos[c]) * audio[c*stride + i]; mix is zeroed first (sdrx.cpp:451) */
void  sdrx_oracle_pan_mix(const float *audio, size_t stride, const int *pos, int n_channels, size_t n, float *mix);
/* float -> S16 with the clamp of sdrx.cpp:649-656 */
void  sdrx_oracle_to_s16(const float *in, size_t n, int16_t *out);

/* ---- channel names -> translator offsets (static functions of sdrx.cpp, restated from the text: PARITY UNPINNED
 * beyond review -- sdrx.cpp cannot be compiled here; integer logic with hand-checked known answers in
 * tests/test_oracle.py) ---- */
/* parse_fq, reference src/sdrx.cpp:957-1021 (0 for an invalid string; aeronautical != 0: channel-name notation) */
uint32_t sdrx_oracle_parse_fq(const char *str, int aeronautical);
/* channel_to_offset, reference src/sdrx.cpp:1449-1478 (name must be a valid aeronautical channel) */
int      sdrx_oracle_channel_to_offset(const char *channel, int32_t tuner_fq);
/* tuner frequency for a set of channels, reference src/sdrx.cpp:1398-1408 */
uint32_t sdrx_oracle_tuner_fq(const char *const *names, size_t n);
/* verify_requested_bandwidth, reference src/sdrx.cpp:1424-1443 (without the override flag) */
int      sdrx_oracle_verify_bandwidth(const char *const *names, size_t n, uint32_t sample_rate_hz);

/* ---- synthetic IQ generator (ours; not a reference function) ---- */
void sdrx_oracle_synth_iq(uint64_t seed, uint64_t n_start, size_t n, float noise_amp,
                          int n_carriers, const float *const *carrier_tables_iq, int N,
                          const float *carrier_amp, uint32_t tri_period, float *out_iq);

#ifdef __cplusplus
}
#endif
#endif

// ref_driver.cpp -- C-ABI wrapper around the REFERENCE's own header-only classes.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes
// msd.hpp / agc.hpp / fir.hpp from the read-only mount (-I$(SDRX_REF)/src, see
// oracle/Makefile) and exposes them through the same C signatures as
// sdrx_oracle.h (prefix sdrx_ref_ instead of sdrx_oracle_), so that tests and
// oracle/gen_fixtures.py can run the restatement and the real thing side by side.
// The outputs go to oracle/_ref/ (git-ignored); nothing here is linked into the
// product library.
//
// Demod (reference src/sdrx.cpp:173-203) lives inside sdrx.cpp, which cannot be
// compiled here (popt/fftw/alsa/sigc++ absent); its AM branch is the single call
// std::abs(sample), which is what sdrx_ref_demod_am does.

#include <complex>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>

#include <msd.hpp>   // reference
#include <agc.hpp>   // reference
#include <fir.hpp>   // reference

extern "C" {

void sdrx_ref_make_translator(int N, int z, int ch_offset, float *out_iq)
{
    // Verbatim expression shape of reference src/sdrx.cpp:1687-1688.
    for (int n = 0; n < N; n++) {
        std::complex<float> e(0.0f, -2.0f * M_PI * n * ch_offset * (float)z/(float)N);
        std::complex<float> t = exp(e);
        out_iq[2 * n] = t.real();
        out_iq[2 * n + 1] = t.imag();
    }
}

struct sdrx_ref_msd { MSD msd; };

sdrx_ref_msd *sdrx_ref_msd_create(const float *translator_iq, size_t n_translator, int n_stages,
                                  const uint32_t *m, const uint32_t *n_taps,
                                  const float *const *taps, int use_ftfir)
{
    std::vector<iqsample_t> translator;
    for (size_t i = 0; i < n_translator; i++)
        translator.emplace_back(translator_iq[2 * i], translator_iq[2 * i + 1]);
    std::vector<MSD::Stage> stages;
    for (int s = 0; s < n_stages; s++)
        stages.push_back(MSD::Stage{m[s], std::vector<float>(taps[s], taps[s] + n_taps[s])});
    return new sdrx_ref_msd{MSD(translator, stages, use_ftfir != 0)};
}

sdrx_ref_msd *sdrx_ref_msd_clone(const sdrx_ref_msd *src) { return new sdrx_ref_msd{src->msd}; }

size_t sdrx_ref_msd_m(const sdrx_ref_msd *h) { return h->msd.m(); }

void sdrx_ref_msd_decimate(sdrx_ref_msd *h, const float *in_iq, size_t in_len, float *out_iq,
                           size_t *out_len)
{
    h->msd.decimate(reinterpret_cast<const iqsample_t *>(in_iq), in_len,
                    reinterpret_cast<iqsample_t *>(out_iq), out_len);
}

void sdrx_ref_msd_destroy(sdrx_ref_msd *h) { delete h; }

// Same POD as sdrx_oracle_agc so the Python side can share one ctypes.Structure;
// the reference objects are kept beside it and the POD mirrors what can be read back.
struct sdrx_ref_agc_pair {
    AGC   agc;
    LfAGC agc_lf;
};

sdrx_ref_agc_pair *sdrx_ref_agc_create_sdrx(int use_lf_agc)
{
    // reference src/sdrx.cpp:1702-1710 on default-constructed members of Channel
    auto *p = new sdrx_ref_agc_pair();
    p->agc.setReference(1.0f);
    p->agc.setAttack(1.0f);
    p->agc.setDecay(0.01f);
    p->agc.setMaxGain(300);
    p->agc_lf.setReference(1.0f);
    p->agc_lf.setAttack(1.0f);
    p->agc_lf.setDecay(0.01f);
    if (use_lf_agc) p->agc_lf.activate();
    return p;
}

void sdrx_ref_agc_adjust(sdrx_ref_agc_pair *p, const float *in_iq, size_t n, float *out_iq)
{
    const iqsample_t *in = reinterpret_cast<const iqsample_t *>(in_iq);
    iqsample_t *out = reinterpret_cast<iqsample_t *>(out_iq);
    for (size_t i = 0; i < n; i++) out[i] = p->agc.adjust(in[i]);
}

void sdrx_ref_lfagc_adjust(sdrx_ref_agc_pair *p, const float *in, size_t n, float *out)
{
    for (size_t i = 0; i < n; i++) out[i] = p->agc_lf.adjust(in[i]);
}

float sdrx_ref_agc_gain(sdrx_ref_agc_pair *p) { return p->agc.gain(); }
float sdrx_ref_lfagc_gain(sdrx_ref_agc_pair *p) { return p->agc_lf.gain(); }
void  sdrx_ref_agc_destroy(sdrx_ref_agc_pair *p) { delete p; }

void sdrx_ref_demod_am(const float *in_iq, size_t n, float *out)
{
    const iqsample_t *in = reinterpret_cast<const iqsample_t *>(in_iq);
    for (size_t i = 0; i < n; i++) out[i] = std::abs(in[i]);
}

struct sdrx_ref_fir { FIR fir; };

sdrx_ref_fir *sdrx_ref_fir_create(const float *c, unsigned n)
{
    return new sdrx_ref_fir{FIR(std::vector<float>(c, c + n))};
}
void sdrx_ref_fir_set_gain(sdrx_ref_fir *f, float gain_db) { f->fir.setGain(gain_db); }
void sdrx_ref_fir_filter(sdrx_ref_fir *f, const float *in, unsigned in_len, float *out)
{
    f->fir.filter(in, in_len, out);
}
void sdrx_ref_fir_destroy(sdrx_ref_fir *f) { delete f; }

struct sdrx_ref_fir2 { FIR2 fir; };

sdrx_ref_fir2 *sdrx_ref_fir2_create(const float *c, unsigned n)
{
    return new sdrx_ref_fir2{FIR2(std::vector<float>(c, c + n))};
}
void sdrx_ref_fir2_set_gain(sdrx_ref_fir2 *f, float gain_db) { f->fir.setGain(gain_db); }
void sdrx_ref_fir2_filter(sdrx_ref_fir2 *f, const float *in, unsigned in_len, float *out)
{
    f->fir.filter(in, in_len, out);
}
void sdrx_ref_fir2_destroy(sdrx_ref_fir2 *f) { delete f; }

// Channel bank = data_cb's loop (sdrx.cpp:342-345) + the AGC/demod/LfAGC lines of
// alsa_write_cb (sdrx.cpp:457-463), squelch taken as open.
struct sdrx_ref_bank {
    std::vector<MSD>   msd;
    std::vector<AGC>   agc;
    std::vector<LfAGC> agc_lf;
    size_t             m_total = 1;
};

sdrx_ref_bank *sdrx_ref_bank_create(int n_channels, const float *const *translators_iq,
                                    const size_t *n_translator, int n_stages, const uint32_t *m,
                                    const uint32_t *n_taps, const float *const *taps,
                                    int use_ftfir, int use_lf_agc)
{
    auto *b = new sdrx_ref_bank();
    std::vector<MSD::Stage> stages;
    for (int s = 0; s < n_stages; s++)
        stages.push_back(MSD::Stage{m[s], std::vector<float>(taps[s], taps[s] + n_taps[s])});
    for (int c = 0; c < n_channels; c++) {
        std::vector<iqsample_t> translator;
        size_t nt = n_translator ? n_translator[c] : 0;
        for (size_t i = 0; i < nt; i++)
            translator.emplace_back(translators_iq[c][2 * i], translators_iq[c][2 * i + 1]);
        b->msd.emplace_back(translator, stages, use_ftfir != 0);
        AGC agc;
        agc.setReference(1.0f);
        agc.setAttack(1.0f);
        agc.setDecay(0.01f);
        agc.setMaxGain(300);
        LfAGC lf;
        lf.setReference(1.0f);
        lf.setAttack(1.0f);
        lf.setDecay(0.01f);
        if (use_lf_agc) lf.activate();
        b->agc.push_back(agc);
        b->agc_lf.push_back(lf);
    }
    if (n_channels) b->m_total = b->msd[0].m();
    return b;
}

void sdrx_ref_bank_process(sdrx_ref_bank *b, const float *in_iq, size_t in_len, float *iq_out,
                           float *audio_out, size_t out_stride, size_t *out_len, int threads)
{
    const int n_ch = (int)b->msd.size();
    const size_t max_out = in_len / b->m_total + 1;
    size_t n_out_first = 0;
    if (threads < 1) threads = 1;
    const iqsample_t *in = reinterpret_cast<const iqsample_t *>(in_iq);

#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
    for (int c = 0; c < n_ch; c++) {
        std::vector<iqsample_t> tmp;
        iqsample_t *dst;
        if (iq_out) {
            dst = reinterpret_cast<iqsample_t *>(iq_out) + (size_t)c * out_stride;
        } else {
            tmp.resize(max_out);
            dst = tmp.data();
        }
        size_t n_out = 0;
        b->msd[c].decimate(in, in_len, dst, &n_out);
        if (audio_out) {
            float *a = audio_out + (size_t)c * out_stride;
            for (size_t i = 0; i < n_out; i++) {
                iqsample_t y = b->agc[c].adjust(dst[i]);
                float s = std::abs(y);
                a[i] = b->agc_lf[c].adjust(s);
            }
        }
        if (c == 0) n_out_first = n_out;
    }
    if (out_len) *out_len = n_out_first;
}

void sdrx_ref_bank_destroy(sdrx_ref_bank *b) { delete b; }

} // extern "C"

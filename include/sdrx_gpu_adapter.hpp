// sdrx_gpu_adapter.hpp -- header-only C++ adapter that keeps sdrx's channel-worker call surface
// (MSD / DS, reference src/msd.hpp:42-75 and src/ds.hpp:32-57) on top of the libsdrxgpu C ABI.
//
// With it, the per-channel loop of data_cb (reference src/sdrx.cpp:327-346) stays as it is:
//
//     for (auto &ch : channels) { ch.msd.decimate(data, data_len, iq_buf_ptr); iq_buf_ptr += 512; }
//     // or, with -t:   ch.ds_ptr->addJob(data, data_len, iq_buf_ptr, latch); ... latch->wait();
//
// but the first call of a block runs ONE batched GPU pass over every channel (sdrxgpu_bank_process)
// and each channel's call then only copies its row out.  The fused IF-AGC -> |.| -> LF-AGC result
// of the same pass (what alsa_write_cb computes per sample, sdrx.cpp:455-463) is available through
// MSD::audio().
//
// Semantics kept from the reference:
//   * MSD(translator, stages, use_ftfir) copies its arguments (msd.hpp:54-68); m() (msd.hpp:71);
//     decimate(in, in_len, out, out_len_ptr) accepts any in_len and is void (msd.hpp:75);
//   * DS(const MSD&) (ds.hpp:34); addJob(data, len, out, latch) counts the latch down when the
//     channel's output is in place (ds.hpp:43, 74);
//   * errors cannot be returned through these signatures: like data_cb (sdrx.cpp:352, 362) the adapter
//     reports on std::cerr and keeps a sticky flag (sdrxgpu::last_error()); outputs are zero-filled.
// Differences (documented in INTEGRATION.md):
//   * all channels constructed before the first decimate()/addJob() of a group form one bank and
//     must be called in lockstep, once per block each (exactly what data_cb does);
//   * a copy of an MSD (Channel assignment, DS's internal copy) is the SAME channel, not a second
//     delay line;
//   * use_ftfir (EXPERIMENTAL in the reference, sdrx.cpp:1199) maps to SDRXGPU_FLAG_FTFIR.
// Optional (one line in main): sdrxgpu::use_devices(8) (devices() = {0, ..., 7}) or sdrxgpu::use_all_devices() BEFORE the first block: the
// channel list of every bank is then cut into contiguous ranges over these GPUs (sdrxgpu_multibank_*: per-device state, the block
// H2D-copied to every device from `in`, every device's rows D2H-copied straight into its slice of the chunk, one worker thread
// per device as DS has one per channel, ds.hpp:43-77).  Same calls, same bits as on one GPU; no RCCL, no second process.
// Optional (one line in main): sdrxgpu::chunk_stride() = CH_IQ_BUF_SIZE promises that the output pointers of one
// block are base + channel_index * stride -- what data_cb does with a ring chunk (iq_buf_ptr += CH_IQ_BUF_SIZE,
// sdrx.cpp:338/344).  The GPU pass then lands directly in the chunk (CRB memory, crb.hpp:110) and the per-channel
// calls copy nothing.
#ifndef SDRX_GPU_ADAPTER_HPP
#define SDRX_GPU_ADAPTER_HPP

#include <atomic>
#include <complex>
#include <cstring>
#include <iostream>
#include <latch>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "sdrx_gpu.h"

namespace sdrxgpu {

using iqsample_t = std::complex<float>;   // reference src/iqsample.hpp:26

namespace detail {
inline std::mutex &config_mutex()
{
    static std::mutex m;
    return m;
}
inline std::string &error_text()
{
    static std::string e;
    return e;
}
inline void set_error(const std::string &what)      // (any thread: a DS worker, the device thread)
{
    std::lock_guard<std::mutex> lock(config_mutex());
    error_text() = what;
}
inline bool &all_devices_flag()
{
    static bool v = false;
    return v;
}
}   // namespace detail

// The sticky error text (empty: none so far).  A copy, taken under a lock: any thread may ask while another one fails.
inline std::string last_error()
{
    std::lock_guard<std::mutex> lock(detail::config_mutex());
    return detail::error_text();
}
inline void clear_last_error() { detail::set_error(std::string()); }

// 0 (default): every decimate()/addJob() copies its channel's row to `out`.  S > 0: see the header comment.  Set once, before the
// first block (atomic, so that a late change is at least not a data race: it takes effect with the next block).
inline std::atomic<size_t> &chunk_stride()
{
    static std::atomic<size_t> v{0};
    return v;
}

// Empty (default): one bank on the current HIP device.  Otherwise the HIP device ordinals the banks created afterwards are cut
// over, in channel order (sdrxgpu_multibank_create); an ordinal may repeat.  Read once per bank, at its first block, under the
// adapter's lock; set it from main before that (use_devices / use_all_devices take the same lock).
inline std::vector<int32_t> &devices()
{
    static std::vector<int32_t> v;
    return v;
}

// devices() = {0, ..., n - 1}; n = 0: back to one bank on the current device; n < 0: nothing changes.  (No HIP header here and no
// probing: an ordinal that does not exist is refused by sdrxgpu_multibank_create at the first block, SDRXGPU_ERR_NO_DEVICE.)
inline void use_devices(int n)
{
    if (n < 0) return;
    std::lock_guard<std::mutex> lock(detail::config_mutex());
    detail::all_devices_flag() = false;
    devices().clear();
    for (int d = 0; d < n; d++) devices().push_back(d);
}

// every GPU the process can see: the C ABI's n_devices == 0 (sdrxgpu_multibank_create(cfg, NULL, 0, ...)) -- the library counts them.
inline void use_all_devices()
{
    std::lock_guard<std::mutex> lock(detail::config_mutex());
    detail::all_devices_flag() = true;
    devices().clear();
}

namespace detail {

struct StageCfg {
    size_t m;
    std::vector<float> h;
    bool operator==(const StageCfg &o) const { return m == o.m && h == o.h; }
};

// One bank = all MSDs constructed with the same stage list before the group's first block.
struct Group {
    std::vector<StageCfg> stages;
    bool use_ftfir = false;
    bool lf_agc = false;
    std::vector<std::vector<iqsample_t>> translators;   // per channel, empty = no mixing
    sdrxgpu_bank *bank = nullptr;
    sdrxgpu_multibank *multi = nullptr;  // instead of `bank` when sdrxgpu::devices() was set at the first block
    bool sealed = false;
    bool failed = false;
    size_t m_total = 1;
    // results of the current block
    std::vector<iqsample_t> iq;      // [n_ch][stride]
    std::vector<float> audio;        // [n_ch][stride]
    size_t stride = 0, out_len = 0;
    iqsample_t *direct_base = nullptr;   // chunk the current block was written into (chunk_stride() mode)
    std::vector<uint64_t> consumed;  // per channel: number of blocks taken
    uint64_t produced = 0;           // number of blocks computed
    std::mutex mu;

    ~Group()
    {
        if (bank) sdrxgpu_bank_destroy(bank);
        if (multi) sdrxgpu_multibank_destroy(multi);
    }

    void fail(const std::string &what)
    {
        failed = true;
        const std::string text = what + ": " + sdrxgpu_last_error();
        set_error(text);
        std::cerr << "Error: sdrxgpu: " << text << std::endl;
    }

    void seal()
    {
        sealed = true;
        consumed.assign(translators.size(), 0);
        std::vector<sdrxgpu_stage> st(stages.size());
        for (size_t i = 0; i < stages.size(); i++) {
            st[i].m = (uint32_t)stages[i].m;
            st[i].n_taps = (uint32_t)stages[i].h.size();
            st[i].taps = stages[i].h.data();
        }
        std::vector<sdrxgpu_channel> ch(translators.size());
        for (size_t c = 0; c < translators.size(); c++) {
            ch[c].translator = reinterpret_cast<const sdrxgpu_iq *>(translators[c].data());
            ch[c].n_translator = (uint32_t)translators[c].size();
        }
        sdrxgpu_bank_cfg cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        cfg.struct_size = sizeof(cfg);
        cfg.flags = (lf_agc ? SDRXGPU_FLAG_LF_AGC_ACTIVE : 0) | (use_ftfir ? SDRXGPU_FLAG_FTFIR : 0);
        cfg.n_stages = (uint32_t)st.size();
        cfg.stages = st.data();
        cfg.n_channels = (uint32_t)ch.size();
        cfg.channels = ch.data();
        sdrxgpu_agc_params_sdrx(&cfg.agc, &cfg.lf_agc);
        cfg.device = -1;
        std::vector<int32_t> devs;
        bool all = false;
        {
            std::lock_guard<std::mutex> lock(config_mutex());
            devs = devices();
            all = all_devices_flag();
        }
        if (all || !devs.empty()) {
            int rc = sdrxgpu_multibank_create(&cfg, all ? nullptr : devs.data(), all ? 0u : (uint32_t)devs.size(), &multi);
            if (rc != SDRXGPU_OK) fail(std::string("sdrxgpu_multibank_create: ") + sdrxgpu_strerror(rc));
            return;
        }
        int rc = sdrxgpu_bank_create(&cfg, &bank);
        if (rc != SDRXGPU_OK) fail(std::string("sdrxgpu_bank_create: ") + sdrxgpu_strerror(rc));
    }

    // channel c wants its share of the block (in, in_len)
    void take(size_t c, const iqsample_t *in, size_t in_len, iqsample_t *out, size_t *out_len_ptr, float *audio_out)
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!sealed) seal();
        if (consumed[c] == produced) {
            // first caller of a new block: one batched pass for everybody
            const size_t need = in_len / m_total + 1;
            const size_t S = chunk_stride();
            const bool direct = S > 0 && out != nullptr && in_len / m_total <= S;
            const size_t want = direct ? S : need;
            if (direct ? stride != S : need > stride) {
                stride = want;
                iq.assign(direct ? 0 : translators.size() * stride, iqsample_t(0.0f, 0.0f));
                audio.assign(translators.size() * stride, 0.0f);
            }
            if (!direct && iq.size() < translators.size() * stride) iq.assign(translators.size() * stride, iqsample_t(0.0f, 0.0f));
            direct_base = direct ? out - c * S : nullptr;
            iqsample_t *dst = direct ? direct_base : iq.data();
            out_len = 0;
            if (!failed) {
                int rc = multi ? sdrxgpu_multibank_process(multi, reinterpret_cast<const sdrxgpu_iq *>(in), in_len,
                                                           reinterpret_cast<sdrxgpu_iq *>(dst), audio.data(), stride, &out_len)
                               : sdrxgpu_bank_process(bank, reinterpret_cast<const sdrxgpu_iq *>(in), in_len,
                                                      reinterpret_cast<sdrxgpu_iq *>(dst), audio.data(), stride, &out_len);
                if (rc != SDRXGPU_OK) fail(std::string(multi ? "sdrxgpu_multibank_process: " : "sdrxgpu_bank_process: ") + sdrxgpu_strerror(rc));
            }
            if (failed) {
                out_len = in_len / m_total;
                if (direct)
                    for (size_t k = 0; k < translators.size(); k++) std::fill(dst + k * stride, dst + k * stride + out_len, iqsample_t(0.0f, 0.0f));
                else
                    std::fill(iq.begin(), iq.end(), iqsample_t(0.0f, 0.0f));
                std::fill(audio.begin(), audio.end(), 0.0f);
            }
            produced++;
        } else if (consumed[c] + 1 != produced) {
            failed = true;
            set_error("channels of one bank must be called in lockstep, once per block");
            std::cerr << "Error: sdrxgpu: channels of one bank must be called in lockstep, once per block" << std::endl;
        }
        consumed[c] = produced;
        if (direct_base && out && out != direct_base + c * stride) {
            // the promise of chunk_stride() does not hold: take a private copy of every row of this block before
            // anything is moved (rows may overlap the places the caller really wants them), serve the rest of the
            // block from it and go back to copying for good
            iq.resize(translators.size() * stride);
            for (size_t k = 0; k < translators.size(); k++)
                std::memcpy(static_cast<void *>(iq.data() + k * stride), direct_base + k * stride, out_len * sizeof(iqsample_t));
            direct_base = nullptr;
            chunk_stride() = 0;
            if (last_error().empty()) {
                set_error("chunk_stride(): an output pointer is not base + channel * stride; copying from now on");
                std::cerr << "Error: sdrxgpu: chunk_stride(): an output pointer is not base + channel * stride; copying from now on" << std::endl;
            }
        }
        if (!direct_base && out) std::memcpy(static_cast<void *>(out), iq.data() + c * stride, out_len * sizeof(iqsample_t));
        if (audio_out) std::memcpy(audio_out, audio.data() + c * stride, out_len * sizeof(float));
        if (out_len_ptr) *out_len_ptr = out_len;
    }
};

inline std::vector<std::shared_ptr<Group>> &groups()
{
    static std::vector<std::shared_ptr<Group>> g;
    return g;
}

inline std::mutex &groups_mutex()
{
    static std::mutex m;
    return m;
}

}  // namespace detail

// --lf-agc for the banks created afterwards (reference: settings.use_lf_agc, sdrx.cpp:1710)
inline bool &default_lf_agc()
{
    static bool v = false;
    return v;
}

class MSD {
public:
    struct Stage {
        size_t m;
        std::vector<float> h;
    };

    MSD() = default;

    MSD(const std::vector<iqsample_t> &translator, const std::vector<MSD::Stage> &stages, bool use_ftfir = false) : m_(1)
    {
        std::vector<detail::StageCfg> st;
        for (const Stage &s : stages) {
            st.push_back({s.m, s.h});
            m_ *= s.m;
        }
        std::lock_guard<std::mutex> lock(detail::groups_mutex());
        for (auto &g : detail::groups())
            if (!g->sealed && g->stages == st && g->use_ftfir == use_ftfir) group_ = g;
        if (!group_) {
            group_ = std::make_shared<detail::Group>();
            group_->stages = st;
            group_->use_ftfir = use_ftfir;
            group_->lf_agc = default_lf_agc();
            group_->m_total = m_;
            detail::groups().push_back(group_);
        }
        slot_ = group_->translators.size();
        group_->translators.push_back(translator);
    }

    size_t m() const { return m_; }

    void decimate(const iqsample_t *in, size_t in_len, iqsample_t *out, size_t *out_len_ptr = nullptr)
    {
        if (!group_) {
            if (out_len_ptr) *out_len_ptr = 0;
            return;
        }
        group_->take(slot_, in, in_len, out, out_len_ptr, nullptr);
    }

    // Extension: the AGC'd AM envelope of the block last taken by decimate() (n = out_len floats).
    const float *audio() const { return group_ ? group_->audio.data() + slot_ * group_->stride : nullptr; }

private:
    size_t m_ = 1;
    std::shared_ptr<detail::Group> group_;
    size_t slot_ = 0;
};

class DS {
public:
    DS(const MSD &msd) : msd_(msd) {}

    void addJob(const iqsample_t *data, unsigned data_len, iqsample_t *out, std::shared_ptr<std::latch> latch)
    {
        msd_.decimate(data, data_len, out);
        latch->count_down();
    }

private:
    MSD msd_;
};

// Destroy every bank (call before the HIP runtime goes away, e.g. at the end of main()).
inline void shutdown()
{
    std::lock_guard<std::mutex> lock(detail::groups_mutex());
    detail::groups().clear();
}

}  // namespace sdrxgpu

#endif  // SDRX_GPU_ADAPTER_HPP

// multibank.hpp -- one channel list over several GPUs of one node, in one process (included at the end of sdrxgpu.hip).
//
// Reference shape being kept: sdrx is ONE process whose device thread calls data_cb once per 32 ms block
// (src/sdrx.cpp:306-364) and whose only parallelism is one independent worker per channel, woken through a job slot and
// joined with a latch (DS, src/ds.hpp:32-78; dispatch src/sdrx.cpp:327-340).  Here the workers are DEVICES: one thread per
// shard holds that shard's bank, takes the block through a job slot (mutex + condition variable, as ds.hpp:43-77 does per
// channel), runs sdrxgpu_bank_process on its channel range and counts the call's latch down.  The block reaches every
// device by that device's own H2D copy from the caller's buffer ("simply 8 independent pinned-host H2D copies"),
// the outputs leave every device straight into its slice of the caller's chunk: no collective, no RCCL, no peer traffic.
#pragma once

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

extern "C" int sdrxgpu_shard_range(uint32_t n_items, uint32_t index, uint32_t n_shards, uint32_t *lo, uint32_t *hi)
{
    if (n_shards == 0 || index >= n_shards || !lo || !hi)
        return fail(SDRXGPU_ERR_INVALID_ARG, "shard_range: shard %u of %u", index, n_shards);
    const uint32_t base = n_items / n_shards, extra = n_items % n_shards;
    *lo = index * base + std::min(index, extra);
    *hi = *lo + base + (index < extra ? 1u : 0u);
    return SDRXGPU_OK;
}

struct sdrxgpu_multibank {
    enum Kind { NONE, PROCESS, PINNED, SYNC, RESET, QUIT };
    struct Job {
        Kind kind = NONE;
        const sdrxgpu_iq *in = nullptr;
        size_t in_len = 0, out_stride = 0;
        sdrxgpu_iq *iq_out = nullptr;           // PROCESS: the caller's whole arrays (a shard writes rows ch_lo ...)
        float *audio_out = nullptr;
        void *const *d_iq_out = nullptr;        // PINNED: one device pointer per shard
        void *const *d_audio_out = nullptr;
    };
    struct Shard {
        sdrxgpu_bank *bank = nullptr;
        int device = 0;
        uint32_t ch_lo = 0, ch_hi = 0;
        std::thread th;
        uint64_t seen = 0;                      // last job sequence number this worker ran
        int rc = SDRXGPU_OK;
        size_t out_len = 0;
        char err[sizeof(g_err)] = "";
    };
    std::vector<Shard> shards;
    uint32_t n_ch = 0, m_total = 1;
    int family = 0;                             // bank_info.matrix_pipe of every shard
    // the job slot
    std::mutex mu;
    std::condition_variable cv_job, cv_done;
    Job job;
    uint64_t seq = 0;
    int outstanding = 0;
    // The same two counters once more as atomics, for a short spin in front of either wait: a condition-variable wake-up is 30 - 60 us
    // on this host, the GPU work of a 1 024-channel shard 120 us -- a stream of blocks posted back to back (bench.py, or a receiver
    // catching up) should not pay one sleep per shard and block.  A thread spins kSpinUs and then sleeps as before: with a block
    // every 32 ms (real time) that is 8 x 50 us of polling per block.
    std::atomic<uint64_t> seq_a{0};
    std::atomic<int> outstanding_a{0};
};
constexpr int kMbSpinUs = 50;

template <class Pred>
static inline void mb_spin(Pred done)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; !done(); i++) {
        if ((i & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(kMbSpinUs)) return;
#if defined(__x86_64__)
        __builtin_ia32_pause();          // (host pass only: the device pass of this file has no such builtin, and never runs this)
#endif
    }
}

static void mb_run_one(sdrxgpu_multibank *mb, sdrxgpu_multibank::Shard &s, size_t index, const sdrxgpu_multibank::Job &j)
{
    (void)mb;
    s.out_len = 0;
    g_err[0] = 0;
    switch (j.kind) {
        case sdrxgpu_multibank::PROCESS:
            s.rc = process_impl(s.bank, j.in, true, j.in_len, j.iq_out ? j.iq_out + (size_t)s.ch_lo * j.out_stride : nullptr,
                                j.audio_out ? j.audio_out + (size_t)s.ch_lo * j.out_stride : nullptr, true, j.out_stride, &s.out_len);
            break;
        case sdrxgpu_multibank::PINNED:
            s.rc = sdrxgpu_bank_process_pinned(s.bank, j.in, j.in_len, j.d_iq_out ? j.d_iq_out[index] : nullptr,
                                               j.d_audio_out ? j.d_audio_out[index] : nullptr, j.out_stride, &s.out_len);
            break;
        case sdrxgpu_multibank::SYNC:
            s.rc = sdrxgpu_bank_synchronize(s.bank);
            break;
        case sdrxgpu_multibank::RESET:
            s.rc = sdrxgpu_bank_reset(s.bank);
            break;
        default:
            s.rc = SDRXGPU_OK;
            break;
    }
    if (s.rc != SDRXGPU_OK) snprintf(s.err, sizeof(s.err), "%s", g_err);      // (g_err is the WORKER's: hand the text to the caller's thread)
}

static void mb_worker(sdrxgpu_multibank *mb, size_t index)
{
    sdrxgpu_multibank::Shard &s = mb->shards[index];
    (void)hipSetDevice(s.device);
    for (;;) {
        sdrxgpu_multibank::Job j;
        mb_spin([&] { return mb->seq_a.load(std::memory_order_acquire) != s.seen; });
        {
            std::unique_lock<std::mutex> lock(mb->mu);
            mb->cv_job.wait(lock, [&] { return mb->seq != s.seen; });
            s.seen = mb->seq;
            j = mb->job;
        }
        if (j.kind != sdrxgpu_multibank::QUIT) mb_run_one(mb, s, index, j);
        {
            std::lock_guard<std::mutex> lock(mb->mu);
            mb->outstanding_a.store(mb->outstanding - 1, std::memory_order_release);
            if (--mb->outstanding == 0) mb->cv_done.notify_all();
        }
        if (j.kind == sdrxgpu_multibank::QUIT) return;
    }
}

// Post one job to every shard and wait for all of them (the latch of ds.hpp:74 / sdrx.cpp:340).  Returns the first failing
// shard's code, its message in the caller's g_err.
static int mb_dispatch(sdrxgpu_multibank *mb, const sdrxgpu_multibank::Job &j, const char *what)
{
    if (mb->shards.size() == 1 && !mb->shards[0].th.joinable()) {
        mb_run_one(mb, mb->shards[0], 0, j);           // one device: no thread, the caller's own
    } else {
        {
            std::unique_lock<std::mutex> lock(mb->mu);
            mb->job = j;
            mb->outstanding = (int)mb->shards.size();
            mb->outstanding_a.store(mb->outstanding, std::memory_order_release);
            mb->seq++;
            mb->seq_a.store(mb->seq, std::memory_order_release);
            mb->cv_job.notify_all();
        }
        mb_spin([&] { return mb->outstanding_a.load(std::memory_order_acquire) == 0; });
        std::unique_lock<std::mutex> lock(mb->mu);
        mb->cv_done.wait(lock, [&] { return mb->outstanding == 0; });
    }
    for (size_t i = 0; i < mb->shards.size(); i++) {
        const sdrxgpu_multibank::Shard &s = mb->shards[i];
        if (s.rc != SDRXGPU_OK) return fail(s.rc, "%s: shard %zu (device %d, channels %u..%u): %s", what, i, s.device, s.ch_lo, s.ch_hi, s.err);
    }
    return SDRXGPU_OK;
}

static int mb_free(sdrxgpu_multibank *mb)
{
    if (!mb) return 0;
    bool threads = false;
    for (auto &s : mb->shards) threads |= s.th.joinable();
    if (threads) {
        {
            std::unique_lock<std::mutex> lock(mb->mu);
            mb->job = sdrxgpu_multibank::Job{};
            mb->job.kind = sdrxgpu_multibank::QUIT;
            mb->outstanding = 0;
            for (auto &s : mb->shards) mb->outstanding += s.th.joinable() ? 1 : 0;
            mb->outstanding_a.store(mb->outstanding, std::memory_order_release);
            mb->seq++;
            mb->seq_a.store(mb->seq, std::memory_order_release);
            mb->cv_job.notify_all();
        }
        for (auto &s : mb->shards)
            if (s.th.joinable()) s.th.join();
    }
    int n_bad = 0;
    for (auto &s : mb->shards) {
        if (!s.bank) continue;
        (void)hipSetDevice(s.device);
        n_bad += free_bank(s.bank);
    }
    delete mb;
    return n_bad;
}

extern "C" int sdrxgpu_multibank_create(const sdrxgpu_bank_cfg *cfg, const int32_t *devices, uint32_t n_devices, sdrxgpu_multibank **out)
{
    if (!cfg || !out) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_create: null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(sdrxgpu_bank_cfg))
        return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_create: struct_size %u != %zu", cfg->struct_size, sizeof(sdrxgpu_bank_cfg));
    if (cfg->n_channels == 0 || !cfg->channels) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_create: no channels");
    if (n_devices && !devices) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_create: %u devices but a null list", n_devices);
    if (cfg->pan_pos)
        return fail(SDRXGPU_ERR_UNSUPPORTED, "multibank_create: the stereo mix-down (pan_pos) sums over ALL channels; it is not available across devices");
    if (cfg->stream) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_create: cfg.stream belongs to one device; a multibank creates its shards' streams");
    if (cfg->flags & SDRXGPU_FLAG_PIPELINED)
        return fail(SDRXGPU_ERR_UNSUPPORTED, "multibank_create: SDRXGPU_FLAG_PIPELINED is for sdrxgpu_bank_process_device only");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(SDRXGPU_ERR_NO_DEVICE, "multibank_create: no HIP device (this library has no CPU path)");
    std::vector<int32_t> devs;
    if (n_devices == 0) {
        for (int d = 0; d < ndev; d++) devs.push_back(d);
    } else {
        devs.assign(devices, devices + n_devices);
    }
    for (int32_t d : devs)
        if (d < 0 || d >= ndev) return fail(SDRXGPU_ERR_NO_DEVICE, "multibank_create: device %d of %d", d, ndev);
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);

    // The one choice whose outputs differ by rounding -- vector kernels, k_stage12m or k_stage123m -- is made for the WHOLE
    // channel list (on the first device: the count of its CUs enters the launch-count model) and pinned in every shard.
    sdrxgpu_bank_cfg whole = *cfg;
    whole.device = devs[0];
    int family = 0;
    int rc = bank_create_guarded(&whole, nullptr, &family);
    if (rc != SDRXGPU_OK) {
        (void)hipSetDevice(prev_dev);
        return rc;
    }
    sdrxgpu_tuning tune = cfg->tuning ? *cfg->tuning : sdrxgpu_tuning{};
    tune.struct_size = sizeof(sdrxgpu_tuning);
    tune.matrix_pipe = family == 2 ? 3u : family == 1 ? 2u : 1u;

    sdrxgpu_multibank *mb = new (std::nothrow) sdrxgpu_multibank();
    if (!mb) return fail(SDRXGPU_ERR_NO_MEM, "multibank_create: host allocation failed");
    mb->n_ch = cfg->n_channels;
    mb->family = family;
    const uint32_t n_sh = (uint32_t)std::min<size_t>(devs.size(), cfg->n_channels);
    try {
        mb->shards.resize(n_sh);
    } catch (const std::bad_alloc &) {
        delete mb;
        return fail(SDRXGPU_ERR_NO_MEM, "multibank_create: host allocation failed");
    }
    for (uint32_t i = 0; i < n_sh; i++) {
        sdrxgpu_multibank::Shard &s = mb->shards[i];
        s.device = devs[i];
        (void)sdrxgpu_shard_range(cfg->n_channels, i, n_sh, &s.ch_lo, &s.ch_hi);
        sdrxgpu_bank_cfg sc = *cfg;
        sc.device = s.device;
        sc.n_channels = s.ch_hi - s.ch_lo;
        sc.channels = cfg->channels + s.ch_lo;
        if (cfg->sql_level) sc.sql_level = cfg->sql_level + s.ch_lo;
        if (cfg->modulation) sc.modulation = cfg->modulation + s.ch_lo;
        // k_tail_agc or k_tail_s + k_agc (same bits): a shard that shares its device with another shard of this list keeps the two
        // short kernels, as two banks on one GPU do (bank_create decides from the banks alive on the device; the first shard of a
        // shared device would otherwise be created "alone" and differ from its neighbours) -- unless the caller's tuning says otherwise
        sdrxgpu_tuning tune_s = tune;
        if (tune_s.tail_agc == 0 && std::count(devs.begin(), devs.begin() + n_sh, s.device) > 1) tune_s.tail_agc = 1;
        sc.tuning = &tune_s;
        rc = bank_create_guarded(&sc, &s.bank, nullptr);
        if (rc != SDRXGPU_OK) {
            char keep[sizeof(g_err)];
            snprintf(keep, sizeof(keep), "%s", g_err);
            mb_free(mb);
            (void)hipSetDevice(prev_dev);
            return fail(rc, "multibank_create: shard %u (device %d, channels %u..%u): %s", i, devs[i], s.ch_lo, s.ch_hi, keep);
        }
        const int fam = s.bank->mfma3 ? 2 : s.bank->mfma ? 1 : 0;
        if (fam != family) {
            mb_free(mb);
            (void)hipSetDevice(prev_dev);
            return fail(SDRXGPU_ERR_HIP, "internal: shard %u runs kernel family %d, the whole list's choice is %d", i, fam, family);
        }
    }
    mb->m_total = mb->shards[0].bank->m_total;
    (void)hipSetDevice(prev_dev);
    if (n_sh > 1) {
        try {
            for (uint32_t i = 0; i < n_sh; i++) mb->shards[i].th = std::thread(mb_worker, mb, (size_t)i);
        } catch (const std::exception &e) {
            mb_free(mb);
            return fail(SDRXGPU_ERR_NO_MEM, "multibank_create: could not start the device threads: %s", e.what());
        }
    }
    *out = mb;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_multibank_destroy(sdrxgpu_multibank *mb)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_destroy: null multibank");
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    const int n_bad = mb_free(mb);
    (void)hipSetDevice(prev_dev);
    if (n_bad) return fail(SDRXGPU_ERR_HIP, "multibank_destroy: %d hipFree call(s) failed", n_bad);
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_multibank_reset(sdrxgpu_multibank *mb)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_reset: null multibank");
    sdrxgpu_multibank::Job j;
    j.kind = sdrxgpu_multibank::RESET;
    return mb_dispatch(mb, j, "multibank_reset");
}

extern "C" int sdrxgpu_multibank_synchronize(sdrxgpu_multibank *mb)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_synchronize: null multibank");
    sdrxgpu_multibank::Job j;
    j.kind = sdrxgpu_multibank::SYNC;
    return mb_dispatch(mb, j, "multibank_synchronize");
}

// every shard stands at the same place of the stream; what one of them would refuse all would: checked here, on the caller's
// thread, before any device is touched (a refused call leaves every shard as it was)
static int mb_validate(sdrxgpu_multibank *mb, const char *what, const sdrxgpu_iq *iq_in, size_t in_len, bool has_outs, size_t out_stride)
{
    if (in_len && !iq_in) return fail(SDRXGPU_ERR_INVALID_ARG, "%s: null input", what);
    for (const auto &s : mb->shards) {
        const int rc = validate_call(s.bank, in_len, has_outs, out_stride);
        if (rc != SDRXGPU_OK) return rc;
    }
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_multibank_process(sdrxgpu_multibank *mb, const sdrxgpu_iq *iq_in, size_t in_len, sdrxgpu_iq *iq_out,
                                         float *audio_out, size_t out_stride, size_t *out_len)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_process: null multibank");
    int rc = mb_validate(mb, "multibank_process", iq_in, in_len, iq_out || audio_out, out_stride);
    if (rc != SDRXGPU_OK) return rc;
    sdrxgpu_multibank::Job j;
    j.kind = sdrxgpu_multibank::PROCESS;
    j.in = iq_in;
    j.in_len = in_len;
    j.iq_out = iq_out;
    j.audio_out = audio_out;
    j.out_stride = out_stride;
    rc = mb_dispatch(mb, j, "multibank_process");
    if (rc == SDRXGPU_OK && out_len) *out_len = mb->shards[0].out_len;
    return rc;
}

extern "C" int sdrxgpu_multibank_process_pinned(sdrxgpu_multibank *mb, const sdrxgpu_iq *iq_in, size_t in_len, void *const *d_iq_out,
                                                void *const *d_audio_out, size_t out_stride, size_t *out_len)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_process_pinned: null multibank");
    int rc = mb_validate(mb, "multibank_process_pinned", iq_in, in_len, d_iq_out || d_audio_out, out_stride);
    if (rc != SDRXGPU_OK) return rc;
    sdrxgpu_multibank::Job j;
    j.kind = sdrxgpu_multibank::PINNED;
    j.in = iq_in;
    j.in_len = in_len;
    j.d_iq_out = d_iq_out;
    j.d_audio_out = d_audio_out;
    j.out_stride = out_stride;
    rc = mb_dispatch(mb, j, "multibank_process_pinned");
    if (rc == SDRXGPU_OK && out_len) *out_len = mb->shards[0].out_len;
    return rc;
}

extern "C" int sdrxgpu_multibank_get_info(sdrxgpu_multibank *mb, sdrxgpu_multibank_info *info)
{
    if (!mb || !info) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_get_info: null argument");
    info->n_shards = (uint32_t)mb->shards.size();
    info->n_channels = mb->n_ch;
    info->m_total = mb->m_total;
    info->matrix_pipe = (uint32_t)mb->family;
    info->samples_in = mb->shards[0].bank->n_in;
    info->samples_out = mb->shards[0].bank->n_out;
    info->device_bytes = 0;
    for (const auto &s : mb->shards) info->device_bytes += s.bank->device_bytes;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_multibank_shard(sdrxgpu_multibank *mb, uint32_t index, sdrxgpu_bank **bank, int32_t *device, uint32_t *ch_lo, uint32_t *ch_hi)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_shard: null multibank");
    if (index >= mb->shards.size()) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_shard: shard %u of %zu", index, mb->shards.size());
    const auto &s = mb->shards[index];
    if (bank) *bank = s.bank;
    if (device) *device = s.device;
    if (ch_lo) *ch_lo = s.ch_lo;
    if (ch_hi) *ch_hi = s.ch_hi;
    return SDRXGPU_OK;
}

extern "C" int sdrxgpu_multibank_get_agc_state(sdrxgpu_multibank *mb, float *agc_gain, float *agc_power, float *lf_gain, float *lf_power)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_get_agc_state: null multibank");
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    int rc = SDRXGPU_OK;
    for (const auto &s : mb->shards) {
        rc = sdrxgpu_bank_get_agc_state(s.bank, agc_gain ? agc_gain + s.ch_lo : nullptr, agc_power ? agc_power + s.ch_lo : nullptr,
                                        lf_gain ? lf_gain + s.ch_lo : nullptr, lf_power ? lf_power + s.ch_lo : nullptr);
        if (rc != SDRXGPU_OK) break;
    }
    (void)hipSetDevice(prev_dev);
    return rc;
}

// what sdrx prints per channel (-v: AGC gains, SNR and squelch state, sdrx.cpp:593-634), over all shards in channel order
extern "C" int sdrxgpu_multibank_get_gains(sdrxgpu_multibank *mb, float *agc_gain, float *lf_agc_gain)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_get_gains: null multibank");
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    int rc = SDRXGPU_OK;
    for (const auto &s : mb->shards)
        if ((rc = sdrxgpu_bank_get_gains(s.bank, agc_gain ? agc_gain + s.ch_lo : nullptr, lf_agc_gain ? lf_agc_gain + s.ch_lo : nullptr))) break;
    (void)hipSetDevice(prev_dev);
    return rc;
}

extern "C" int sdrxgpu_multibank_get_squelch(sdrxgpu_multibank *mb, float *snr_db, unsigned char *state)
{
    if (!mb) return fail(SDRXGPU_ERR_INVALID_ARG, "multibank_get_squelch: null multibank");
    int prev_dev = 0;
    (void)hipGetDevice(&prev_dev);
    int rc = SDRXGPU_OK;
    for (const auto &s : mb->shards)
        if ((rc = sdrxgpu_bank_get_squelch(s.bank, snr_db ? snr_db + s.ch_lo : nullptr, state ? state + s.ch_lo : nullptr))) break;
    (void)hipSetDevice(prev_dev);
    return rc;
}

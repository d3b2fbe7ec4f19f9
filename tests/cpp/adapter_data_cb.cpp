// adapter_data_cb.cpp -- drives include/sdrx_gpu_adapter.hpp exactly the way sdrx's data_cb drives
// MSD / DS (reference src/sdrx.cpp:327-346, set-up 1680-1714) and checks every output against the
// oracle (oracle/sdrx_oracle.h).  Test program: built and run by tests/test_gpu_adapter.py on the GPU box.
//
//   adapter_data_cb <taps.txt> <n_channels> <n_blocks> <serial|ds|pool|serial_direct|ds_direct|pool_direct|broken_direct>
//
// pool: the USE_THREAD_POOL build of sdrx (CMakeLists.txt:104-113): data_cb submits one lambda per channel to a pool of
// worker threads (src/sdrx.cpp:331-335, the pool is src/tp.hpp:31-78) -- here a 4-thread pool of this test's own -- and
// waits on the latch.  Its outputs must equal the serial loop's bit for bit.
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <latch>
#include <memory>
#include <mutex>
#include <queue>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdrx_gpu_adapter.hpp"
#include "../../oracle/sdrx_oracle.h"

using sdrxgpu::iqsample_t;

#define CH_IQ_BUF_SIZE 512   // reference src/sdrx.cpp:129

// A pool of worker threads taking std::function jobs from one queue: the shape of the reference's ThreadPool
// (submit() + N workers, src/tp.hpp:31-78), written for this test.
class WorkerPool {
public:
    explicit WorkerPool(unsigned n)
    {
        for (unsigned i = 0; i < n; i++) workers_.emplace_back([this] { run(); });
    }
    ~WorkerPool()
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            done_ = true;
        }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void submit(std::function<void()> job)
    {
        {
            std::lock_guard<std::mutex> l(mu_);
            jobs_.push(std::move(job));
        }
        cv_.notify_one();
    }

private:
    void run()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> l(mu_);
                cv_.wait(l, [this] { return done_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                job = std::move(jobs_.front());
                jobs_.pop();
            }
            job();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::queue<std::function<void()>> jobs_;
    std::vector<std::thread> workers_;
    bool done_ = false;
};

struct Channel {            // the hot-path members of sdrx's Channel (sdrx.cpp:207-234)
    int ch_offset;
    sdrxgpu::MSD msd;
    sdrxgpu::DS *ds_ptr = nullptr;
};

static double rel_err(const float *got, const float *ref, size_t n_cplx_or_real, int width)
{
    double rms = 0;
    const size_t n = n_cplx_or_real;
    for (size_t i = 0; i < n; i++) {
        double m2 = 0;
        for (int k = 0; k < width; k++) m2 += (double)ref[width * i + k] * ref[width * i + k];
        rms += m2;
    }
    rms = std::sqrt(rms / (n ? n : 1));
    if (rms == 0) rms = 1;
    double worst = 0;
    for (size_t i = 0; i < n; i++) {
        double d2 = 0, m2 = 0;
        for (int k = 0; k < width; k++) {
            const double d = (double)got[width * i + k] - ref[width * i + k];
            d2 += d * d;
            m2 += (double)ref[width * i + k] * ref[width * i + k];
        }
        const double den = std::max(std::sqrt(m2), rms);
        worst = std::max(worst, std::sqrt(d2) / den);
    }
    return worst;
}

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int n_ch = std::atoi(argv[2]), n_blocks = std::atoi(argv[3]);
    const std::string mode = argv[4];
    const bool use_threaded_ds = mode == "ds" || mode == "ds_direct";
    const bool use_pool = mode == "pool" || mode == "pool_direct";
    const bool direct = mode.find("direct") != std::string::npos;
    const bool broken = mode == "broken_direct";      // promises a chunk layout, then hands out other pointers
    if (direct) sdrxgpu::chunk_stride() = CH_IQ_BUF_SIZE;
    // optional: the bank cut over this many shards (all on device 0 -- the box has one GPU): sdrxgpu::devices(), the one line a
    // multi-GPU sdrx adds to main()
    const int n_shards = argc > 5 ? std::atoi(argv[5]) : 0;
    if (n_shards > 0) sdrxgpu::devices() = std::vector<int32_t>((size_t)n_shards, 0);
    // n_shards < 0: sdrxgpu::use_all_devices() -- every GPU the process sees (the C ABI's n_devices == 0), however many those are
    if (n_shards < 0) sdrxgpu::use_all_devices();
    const size_t pitch = broken ? CH_IQ_BUF_SIZE + 8 : CH_IQ_BUF_SIZE;

    // stage list (what sdrx.cpp:1660-1668 builds for 10 MS/s), read from the file the test wrote
    std::ifstream f(argv[1]);
    int N, z, n_stages, block_len;
    f >> N >> z >> block_len >> n_stages;
    std::vector<sdrxgpu::MSD::Stage> stages(n_stages);
    for (auto &s : stages) {
        size_t nt;
        f >> s.m >> nt;
        s.h.resize(nt);
        for (auto &h : s.h) {
            std::string hex;
            f >> hex;
            uint32_t u = (uint32_t)std::stoul(hex, nullptr, 16);
            std::memcpy(&h, &u, 4);
        }
    }

    // --- set-up, as sdrx.cpp:1680-1699 ---
    std::vector<Channel> channels(n_ch);
    std::vector<std::vector<iqsample_t>> tables(n_ch);
    for (int c = 0; c < n_ch; c++) {
        Channel &ch = channels[c];
        ch.ch_offset = c - n_ch / 2;                       // includes 0: empty translator (sdrx.cpp:1685)
        std::vector<iqsample_t> translator;
        if (ch.ch_offset != 0) {
            translator.resize(N);
            sdrxgpu_make_translator(N, z, ch.ch_offset, reinterpret_cast<sdrxgpu_iq *>(translator.data()));
        }
        tables[c] = translator;
        ch.msd = sdrxgpu::MSD(translator, stages, false);
        if (use_threaded_ds) ch.ds_ptr = new sdrxgpu::DS(ch.msd);
    }

    // oracle bank over the same channels
    std::vector<const float *> tptr(n_ch);
    std::vector<size_t> tlen(n_ch);
    std::vector<uint32_t> m(n_stages), nt(n_stages);
    std::vector<const float *> taps(n_stages);
    for (int c = 0; c < n_ch; c++) {
        tptr[c] = reinterpret_cast<const float *>(tables[c].data());
        tlen[c] = tables[c].size();
    }
    for (int s = 0; s < n_stages; s++) {
        m[s] = (uint32_t)stages[s].m;
        nt[s] = (uint32_t)stages[s].h.size();
        taps[s] = stages[s].h.data();
    }
    sdrx_oracle_bank *ob = sdrx_oracle_bank_create(n_ch, tptr.data(), tlen.data(), n_stages, m.data(), nt.data(), taps.data(), 0, 0);

    // pool modes: the same channels once more as a second bank, driven by the plain serial loop, for the bitwise comparison
    // (an empty call on every channel first: it closes the first bank, so that the MSDs built next form a bank of their own)
    std::vector<Channel> serial_channels(use_pool ? n_ch : 0);
    if (use_pool)
        for (auto &ch : channels) ch.msd.decimate(nullptr, 0, nullptr);
    // (... and that second bank on ONE device: pool_equals_serial is then also "cut over devices == not cut", bit for bit)
    if (use_pool) sdrxgpu::use_devices(0);
    for (int c = 0; c < (int)serial_channels.size(); c++) serial_channels[c].msd = sdrxgpu::MSD(tables[c], stages, false);
    std::vector<iqsample_t> serial_iq((size_t)serial_channels.size() * CH_IQ_BUF_SIZE);
    std::unique_ptr<WorkerPool> pool(use_pool ? new WorkerPool(4) : nullptr);
    bool pool_equals_serial = true;

    std::vector<iqsample_t> data(block_len), ring((size_t)2 * n_ch * (CH_IQ_BUF_SIZE + 8)), ref_iq((size_t)n_ch * CH_IQ_BUF_SIZE);
    std::vector<float> ref_audio((size_t)n_ch * CH_IQ_BUF_SIZE);
    double worst_iq = 0, worst_audio = 0;
    for (int blk = 0; blk < n_blocks; blk++) {
        sdrx_oracle_synth_iq(4242, (uint64_t)blk * block_len, block_len, 0.25f, 0, nullptr, 1, nullptr, 10000,
                             reinterpret_cast<float *>(data.data()));
        const unsigned data_len = (unsigned)block_len;

        // --- the loop of data_cb, sdrx.cpp:327-346, verbatim in shape ---
        iqsample_t *const chunk = ring.data() + (size_t)(blk & 1) * n_ch * (CH_IQ_BUF_SIZE + 8);   // chunks alternate, like a ring
        iqsample_t *iq_buf_ptr = chunk;
        if (use_pool) {
            for (size_t c = 0; c < serial_channels.size(); c++)
                serial_channels[c].msd.decimate(data.data(), data_len, serial_iq.data() + c * CH_IQ_BUF_SIZE);
            auto latch = std::make_shared<std::latch>(channels.size());
            for (auto &ch : channels) {
                const iqsample_t *d = data.data();
                pool->submit([&ch, d, data_len, out = iq_buf_ptr, latch]() {      // sdrx.cpp:331-335
                    ch.msd.decimate(d, data_len, out);
                    latch->count_down();
                });
                iq_buf_ptr += pitch;
            }
            latch->wait();
            for (int c = 0; c < n_ch; c++)
                if (std::memcmp(chunk + (size_t)c * pitch, serial_iq.data() + (size_t)c * CH_IQ_BUF_SIZE, CH_IQ_BUF_SIZE * sizeof(iqsample_t)) != 0 ||
                    std::memcmp(channels[c].msd.audio(), serial_channels[c].msd.audio(), CH_IQ_BUF_SIZE * sizeof(float)) != 0)
                    pool_equals_serial = false;
        } else if (use_threaded_ds) {
            auto latch = std::make_shared<std::latch>(channels.size());
            for (auto &ch : channels) {
                ch.ds_ptr->addJob(data.data(), data_len, iq_buf_ptr, latch);
                iq_buf_ptr += pitch;
            }
            latch->wait();
        } else {
            for (auto &ch : channels) {
                ch.msd.decimate(data.data(), data_len, iq_buf_ptr);
                iq_buf_ptr += pitch;
            }
        }

        size_t n_out = 0;
        sdrx_oracle_bank_process(ob, reinterpret_cast<const float *>(data.data()), block_len, reinterpret_cast<float *>(ref_iq.data()),
                                 ref_audio.data(), CH_IQ_BUF_SIZE, &n_out, 4);
        if (n_out != CH_IQ_BUF_SIZE) {
            std::printf("FAIL: oracle produced %zu outputs\n", n_out);
            return 1;
        }
        for (int c = 0; c < n_ch; c++) {
            worst_iq = std::max(worst_iq, rel_err(reinterpret_cast<const float *>(chunk + (size_t)c * pitch),
                                                  reinterpret_cast<const float *>(ref_iq.data() + (size_t)c * CH_IQ_BUF_SIZE), CH_IQ_BUF_SIZE, 2));
            worst_audio = std::max(worst_audio, rel_err(channels[c].msd.audio(), ref_audio.data() + (size_t)c * CH_IQ_BUF_SIZE, CH_IQ_BUF_SIZE, 1));
        }
    }
    for (auto &ch : channels) delete ch.ds_ptr;
    sdrx_oracle_bank_destroy(ob);
    // a broken promise is reported (sticky error) and the rows still end up where the caller asked for them
    const bool err_ok = broken ? !sdrxgpu::last_error().empty() : sdrxgpu::last_error().empty();
    const bool ok = err_ok && worst_iq <= 1e-5 && worst_audio <= 1e-5 && pool_equals_serial;
    std::printf("%s: mode=%s channels=%d blocks=%d shards=%d rel_err iq=%.3e audio=%.3e %s\n", ok ? "PASS" : "FAIL", argv[4], n_ch, n_blocks,
                n_shards, worst_iq, worst_audio, sdrxgpu::last_error().c_str());
    pool.reset();
    channels.clear();
    serial_channels.clear();
    sdrxgpu::shutdown();
    return ok ? 0 : 1;
}

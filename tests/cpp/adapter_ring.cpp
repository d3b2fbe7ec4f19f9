// adapter_ring.cpp -- the zero-copy path behind a chunk ring, the way sdrx runs it: data_cb acquires a chunk of the ring,
// lets every channel's MSD write its 512 outputs into it and commits it (reference src/sdrx.cpp:324-351); alsa_worker, a chunk
// behind on another thread, acquires it for reading and commits the read (src/sdrx.cpp:440, 638); when the ring is full the
// block is dropped whole (src/sdrx.cpp:360-363).  Ring semantics restated from the reference's CRB (src/crb.hpp:64-189:
// acquireWrite / commitWrite / acquireRead / commitRead, num_chunks + 1 slots, single producer, single consumer) and its
// test's wrap-around and overrun cases (src/crb_test.cpp:189-352) -- an own, minimal implementation: the reference does not
// travel to the GPU box.
//
// Here the chunks are PAGE-LOCKED (sdrxgpu_host_register, what INTEGRATION.md tells a maintainer to do with the CRB's
// chunks) and the adapter runs in chunk_stride() mode: the GPU pass lands directly in the acquired chunk.  Checked, for 24
// blocks over an 8-chunk ring (three wraps) with the consumer one chunk behind and stalled for a while in the middle:
//   * every committed chunk equals the oracle's output for the block it carries (<= 1e-5);
//   * no chunk changes after its commit (bitwise against a snapshot taken at commit, compared when the consumer reads it
//     one or more blocks later -- a late D2H would show here);
//   * while the consumer stalls the ring fills, acquireWrite fails, the block is dropped (neither the bank nor the oracle sees
//     it) and the stream continues correctly afterwards.
//
//   adapter_ring <taps.txt> <n_channels>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sdrx_gpu_adapter.hpp"
#include "../../oracle/sdrx_oracle.h"

using sdrxgpu::iqsample_t;

#define CH_IQ_BUF_SIZE 512   // reference src/sdrx.cpp:129

struct Metadata {            // what data_cb stores beside a chunk (src/sdrx.cpp:147, 348-350), plus the test's sequence number
    uint64_t seq;
    float pwr_dbfs;
};

// single-producer / single-consumer ring of fixed-size chunks, 64-byte aligned (crb.hpp:56), num_chunks usable slots
template <typename T, typename M>
class ChunkRing {
public:
    ChunkRing(size_t chunk_size, size_t num_chunks) : n_(num_chunks + 1), chunk_size_(chunk_size), bufs_(n_), meta_(n_)
    {
        for (auto &b : bufs_) b = static_cast<T *>(::operator new[](chunk_size * sizeof(T), std::align_val_t{64}));
    }
    ~ChunkRing()
    {
        for (auto b : bufs_) ::operator delete[](b, std::align_val_t{64});
    }
    bool acquireWrite(T **buf, M **m)
    {
        const size_t w = w_.load(std::memory_order_relaxed), r = r_.load(std::memory_order_acquire);
        if ((w + 1) % n_ == r) return false;        // full: one slot always stays free
        *buf = bufs_[w];
        *m = &meta_[w];
        acquired_w_ = true;
        return true;
    }
    bool commitWrite()
    {
        if (!acquired_w_) return false;             // spurious commit (crb_test.cpp:107)
        acquired_w_ = false;
        w_.store((w_.load(std::memory_order_relaxed) + 1) % n_, std::memory_order_release);
        return true;
    }
    bool acquireRead(const T **buf, const M **m)
    {
        const size_t r = r_.load(std::memory_order_relaxed), w = w_.load(std::memory_order_acquire);
        if (r == w) return false;                   // empty (crb_test.cpp:99)
        *buf = bufs_[r];
        *m = &meta_[r];
        acquired_r_ = true;
        return true;
    }
    bool commitRead()
    {
        if (!acquired_r_) return false;
        acquired_r_ = false;
        r_.store((r_.load(std::memory_order_relaxed) + 1) % n_, std::memory_order_release);
        return true;
    }
    size_t available() const { return (w_.load(std::memory_order_acquire) + n_ - r_.load(std::memory_order_acquire)) % n_; }
    T *chunk(size_t i) { return bufs_[i]; }
    size_t slots() const { return n_; }
    size_t chunk_bytes() const { return chunk_size_ * sizeof(T); }

private:
    const size_t n_, chunk_size_;
    std::vector<T *> bufs_;
    std::vector<M> meta_;
    std::atomic<size_t> w_{0}, r_{0};
    bool acquired_w_ = false, acquired_r_ = false;
};

static double rel_err(const float *got, const float *ref, size_t n, int width)
{
    double rms = 0;
    for (size_t i = 0; i < n * width; i++) rms += (double)ref[i] * ref[i];
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
        worst = std::max(worst, std::sqrt(d2) / std::max(std::sqrt(m2), rms));
    }
    return worst;
}

struct Expected {           // what the consumer must find in the chunk with this sequence number
    uint64_t seq;
    std::vector<iqsample_t> oracle_iq;    // the oracle's outputs for the block
    std::vector<iqsample_t> at_commit;    // the chunk's bytes when it was committed
};

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    const int n_ch = std::atoi(argv[2]);
    const int n_blocks = 24, n_chunks = 8;
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

    // ---- set-up as sdrx.cpp:1680-1699, 1742: channels, ring of 8 chunks of 512 * n_ch samples, page-locked ----
    sdrxgpu::chunk_stride() = CH_IQ_BUF_SIZE;
    std::vector<std::vector<iqsample_t>> tables(n_ch);
    std::vector<sdrxgpu::MSD> msd(n_ch);
    for (int c = 0; c < n_ch; c++) {
        const int off = c - n_ch / 2;
        if (off != 0) {
            tables[c].resize(N);
            sdrxgpu_make_translator(N, z, off, reinterpret_cast<sdrxgpu_iq *>(tables[c].data()));
        }
        msd[c] = sdrxgpu::MSD(tables[c], stages, false);
    }
    const size_t chunk_size = (size_t)CH_IQ_BUF_SIZE * n_ch;
    ChunkRing<iqsample_t, Metadata> rb(chunk_size, n_chunks);
    for (size_t i = 0; i < rb.slots(); i++)
        if (sdrxgpu_host_register(rb.chunk(i), rb.chunk_bytes()) != 0) {
            std::printf("FAIL: sdrxgpu_host_register: %s\n", sdrxgpu_last_error());
            return 1;
        }

    std::vector<const float *> tptr(n_ch), taps(n_stages);
    std::vector<size_t> tlen(n_ch);
    std::vector<uint32_t> m(n_stages), nt(n_stages);
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

    std::mutex mu;
    std::deque<Expected> expected;
    std::atomic<bool> producer_done{false}, stall{false};
    std::atomic<int> chunks_read{0}, chunks_bad_oracle{0}, chunks_changed{0}, seq_errors{0};
    double worst_iq = 0;

    // ---- the consumer (alsa_worker's side, sdrx.cpp:440 / 638): one chunk behind the producer ----
    std::thread consumer([&] {
        uint64_t next_seq = 0;
        for (;;) {
            const bool done = producer_done.load();
            if (stall.load() || rb.available() < (done ? 1u : 2u)) {
                if (done && rb.available() == 0) break;
                std::this_thread::sleep_for(std::chrono::microseconds(200));
                continue;
            }
            const iqsample_t *buf;
            const Metadata *md;
            if (!rb.acquireRead(&buf, &md)) continue;
            Expected e;
            {
                std::lock_guard<std::mutex> l(mu);
                e = std::move(expected.front());
                expected.pop_front();
            }
            if (md->seq != e.seq || e.seq < next_seq) seq_errors++;
            next_seq = e.seq + 1;
            if (std::memcmp(buf, e.at_commit.data(), chunk_size * sizeof(iqsample_t)) != 0) chunks_changed++;
            double w = 0;
            for (int c = 0; c < n_ch; c++)
                w = std::max(w, rel_err(reinterpret_cast<const float *>(buf + (size_t)c * CH_IQ_BUF_SIZE),
                                        reinterpret_cast<const float *>(e.oracle_iq.data() + (size_t)c * CH_IQ_BUF_SIZE), CH_IQ_BUF_SIZE, 2));
            worst_iq = std::max(worst_iq, w);
            if (w > 1e-5) chunks_bad_oracle++;
            rb.commitRead();
            chunks_read++;
        }
    });

    // ---- the producer: data_cb, sdrx.cpp:306-364 ----
    std::vector<iqsample_t> data(block_len), ref_iq(chunk_size);
    std::vector<float> ref_audio((size_t)n_ch * CH_IQ_BUF_SIZE);
    int overruns = 0, committed = 0;
    bool spurious_commit_refused = !rb.commitWrite();          // (crb_test.cpp:107: a commit without an acquire is refused)
    for (int blk = 0; blk < n_blocks; blk++) {
        if (blk == 5) stall = true;                            // the consumer falls behind: the ring fills ...
        if (blk == 17) stall = false;                          // ... and drains again
        sdrx_oracle_synth_iq(777, (uint64_t)blk * block_len, block_len, 0.25f, 0, nullptr, 1, nullptr, 10000, reinterpret_cast<float *>(data.data()));
        iqsample_t *iq_buf_ptr;
        Metadata *md;
        if (!rb.acquireWrite(&iq_buf_ptr, &md)) {              // "Ring buffer full": the block is dropped whole (sdrx.cpp:360-363)
            overruns++;
            continue;
        }
        iqsample_t *const chunk = iq_buf_ptr;
        for (auto &ch : msd) {                                 // sdrx.cpp:342-345
            ch.decimate(data.data(), (unsigned)block_len, iq_buf_ptr);
            iq_buf_ptr += CH_IQ_BUF_SIZE;
        }
        size_t n_out = 0;
        sdrx_oracle_bank_process(ob, reinterpret_cast<const float *>(data.data()), block_len, reinterpret_cast<float *>(ref_iq.data()),
                                 ref_audio.data(), CH_IQ_BUF_SIZE, &n_out, 4);
        md->seq = (uint64_t)blk;
        md->pwr_dbfs = 0.0f;
        {
            std::lock_guard<std::mutex> l(mu);
            expected.push_back(Expected{(uint64_t)blk, ref_iq, std::vector<iqsample_t>(chunk, chunk + chunk_size)});
        }
        rb.commitWrite();
        committed++;
        if (stall.load() == false) std::this_thread::sleep_for(std::chrono::milliseconds(2));     // (lets the consumer follow, one chunk behind)
    }
    producer_done = true;
    consumer.join();

    for (size_t i = 0; i < rb.slots(); i++) sdrxgpu_host_unregister(rb.chunk(i));
    sdrx_oracle_bank_destroy(ob);
    const bool ok = spurious_commit_refused && chunks_read.load() == committed && committed + overruns == n_blocks && overruns >= 1 &&
                    committed >= 2 * n_chunks && chunks_bad_oracle.load() == 0 && chunks_changed.load() == 0 && seq_errors.load() == 0 &&
                    sdrxgpu::last_error().empty();
    std::printf("%s: ring of %d page-locked chunks, %d blocks: %d committed and read, %d dropped on a full ring, %d changed after commit, "
                "%d off the oracle (worst %.3e), %d out of sequence %s\n", ok ? "PASS" : "FAIL", n_chunks, n_blocks, committed, overruns,
                chunks_changed.load(), chunks_bad_oracle.load(), worst_iq, seq_errors.load(), sdrxgpu::last_error().c_str());
    msd.clear();
    sdrxgpu::shutdown();
    return ok ? 0 : 1;
}

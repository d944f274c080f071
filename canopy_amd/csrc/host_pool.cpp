// host_pool.cpp -- the host side of "a new probability vector": log(p) of E values, on persistent worker threads.
//
// The canonical logp of the path is glibc's log evaluated on the host (include/canopy.h; the oracle restates the same
// call), so a new BasicEvent::p() vector (reference src/mef/openpsa/event/event.h:93-96) costs E calls of log before
// anything can be uploaded: ~1.5 ms on one core at E = 1e5, four times the 0.33 ms pass it feeds. The workers below take
// that to well under 0.1 ms; they are started by the first call that has enough values to share and are never joined (a
// leaked singleton: nothing runs at process exit). A sequence of quantifications calls in here every few hundred
// microseconds: a worker that has just finished a job keeps polling for the next one for a short while before it goes to
// sleep on the condition variable (a sleeping thread takes 50-100 us to come back, as much as the whole job), and the caller
// polls for completion instead of sleeping.
#include <math.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "canopy_internal.h"

namespace canopy {

namespace {

inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

constexpr auto kSpinBeforeSleep = std::chrono::microseconds(400); // longer than the gap between two passes of a large model

struct Pool {
    std::mutex mu;
    std::condition_variable cv_work;
    std::vector<std::thread> workers;
    std::mutex gate; // one job at a time
    // the job in flight
    const std::function<void(int64_t, int64_t)>* fn = nullptr;
    int64_t n = 0, grain = 1;
    std::atomic<int64_t> next{0};
    std::atomic<uint64_t> generation{0};
    std::atomic<int> running{0};
    std::atomic<int> sleeping{0};

    void drain() { // take grains until none is left (workers and the caller alike)
        for (;;) {
            const int64_t lo = next.fetch_add(grain, std::memory_order_relaxed);
            if (lo >= n) return;
            const int64_t hi = lo + grain < n ? lo + grain : n;
            (*fn)(lo, hi);
        }
    }

    void worker() {
        uint64_t seen = 0;
        for (;;) {
            // poll for a while, then sleep
            const auto t0 = std::chrono::steady_clock::now();
            bool got = false;
            while (std::chrono::steady_clock::now() - t0 < kSpinBeforeSleep) {
                if (generation.load(std::memory_order_acquire) != seen) { got = true; break; }
                cpu_relax();
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(mu);
                sleeping.fetch_add(1, std::memory_order_relaxed);
                cv_work.wait(lk, [&] { return generation.load(std::memory_order_acquire) != seen; });
                sleeping.fetch_sub(1, std::memory_order_relaxed);
            }
            seen = generation.load(std::memory_order_acquire);
            drain();
            running.fetch_sub(1, std::memory_order_acq_rel);
        }
    }

    void start(int n_workers) {
        for (int i = (int)workers.size(); i < n_workers; ++i) {
            workers.emplace_back([this] { worker(); });
            workers.back().detach();
        }
    }
};

Pool* pool() {
    static Pool* p = new Pool(); // leaked on purpose: detached workers may outlive static destruction
    return p;
}

int pool_width() {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    unsigned w = hw / 2; // leave the other half to the caller's own threads (one process per GPU, 16+ cores each)
    if (w > 15) w = 15;
    return (int)w;       // workers beside the calling thread
}

} // namespace

void host_parallel_for(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& fn) {
    if (n <= 0) return;
    if (grain < 1) grain = 1;
    const int width = pool_width();
    if (width == 0 || n <= 2 * grain) { // not worth a wake-up
        fn(0, n);
        return;
    }
    Pool* p = pool();
    std::lock_guard<std::mutex> one_job(p->gate);
    {
        std::lock_guard<std::mutex> lk(p->mu);
        p->start(width);
        p->fn = &fn;
        p->n = n;
        p->grain = grain;
        p->next.store(0, std::memory_order_relaxed);
        p->running.store((int)p->workers.size(), std::memory_order_relaxed);
        p->generation.fetch_add(1, std::memory_order_release);
    }
    if (p->sleeping.load(std::memory_order_relaxed) > 0) p->cv_work.notify_all();
    p->drain(); // the caller works too
    while (p->running.load(std::memory_order_acquire) != 0) cpu_relax(); // (the job is tens of microseconds long)
    p->fn = nullptr;
}

int64_t host_log_values(const double* p, double* out, int64_t n, bool is_log) {
    std::atomic<int64_t> bad{-1};
    auto body = [&](int64_t lo, int64_t hi) {
        int64_t first_bad = -1;
        for (int64_t i = lo; i < hi; ++i) {
            const double v = p[i];
            if (is_log) {
                if (!(v <= 0.0) && first_bad < 0) first_bad = i; // NaN or a log-probability above 0
                out[i] = v;
            } else {
                if (!(v >= 0.0 && v <= 1.0) && first_bad < 0) first_bad = i;
                out[i] = log(v); // glibc log on the host: the canonical logp
            }
        }
        if (first_bad >= 0) {
            int64_t cur = bad.load();
            while ((cur < 0 || first_bad < cur) && !bad.compare_exchange_weak(cur, first_bad)) {}
        }
    };
    host_parallel_for(n, 4096, body);
    return bad.load();
}

} // namespace canopy

// host_pool.cpp -- the host side of "a new probability vector": log(p) of E values, on persistent worker threads.
//
// The canonical logp of the path is glibc's log evaluated on the host (include/canopy.h; the oracle restates the same
// call), so a new BasicEvent::p() vector (reference src/mef/openpsa/event/event.h:93-96) costs E calls of log before
// anything can be uploaded: ~1.5 ms on one core at E = 1e5, four times the 0.34 ms pass it feeds. The workers below take
// that to ~0.1 ms; they are started by the first call that has enough values to share, sleep on a condition variable
// between jobs and are never joined (a leaked singleton: nothing runs at process exit).
#include <math.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "canopy_internal.h"

namespace canopy {

namespace {

struct Pool {
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> workers;
    // the one job in flight (callers are serialised by `gate`)
    std::mutex gate;
    const std::function<void(int64_t, int64_t)>* fn = nullptr;
    int64_t n = 0, grain = 1;
    std::atomic<int64_t> next{0};
    uint64_t generation = 0;
    int running = 0;

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
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return generation != seen; });
            seen = generation;
            lk.unlock();
            drain();
            lk.lock();
            if (--running == 0) cv_done.notify_one();
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
        p->running = (int)p->workers.size();
        ++p->generation;
    }
    p->cv_work.notify_all();
    p->drain(); // the caller works too
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->running == 0; });
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

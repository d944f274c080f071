// TSan driver for canopy_amd/csrc/host_pool.cpp (four concurrent callers, workers going to sleep and waking up):
//   hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -Xarch_host -fsanitize=thread -Iinclude -Icanopy_amd/csrc -c canopy_amd/csrc/host_pool.cpp -o /tmp/host_pool.o
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -g -std=c++17 -fsanitize=thread benchmarks/sanitize/host_pool_tsan.cpp /tmp/host_pool.o -o /tmp/host_pool_tsan -lpthread && /tmp/host_pool_tsan
#include <cmath>
#include <cstdio>
#include <thread>
#include <vector>
#include <functional>
#include <cstdint>
namespace canopy {
void host_parallel_for(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& fn);
int64_t host_log_values(const double* p, double* out, int64_t n, bool is_log);
}
int main() {
    const int64_t n = 100000;
    std::vector<std::thread> th;
    int bad = 0;
    for (int t = 0; t < 4; ++t) th.emplace_back([&, t] {
        std::vector<double> p(n), out(n);
        for (int64_t i = 0; i < n; ++i) p[i] = (double)(i + 1 + t) / (double)(n + 8);
        for (int rep = 0; rep < 300; ++rep) {
            if (rep % 50 == 49) std::this_thread::sleep_for(std::chrono::milliseconds(2)); // let the workers go to sleep
            const int64_t r = canopy::host_log_values(p.data(), out.data(), n, false);
            if (r != -1) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
            for (int64_t i = 0; i < n; i += 997) if (out[i] != std::log(p[i])) __atomic_fetch_add(&bad, 1, __ATOMIC_RELAXED);
            out[(rep * 31) % n] = 0.0;
        }
    });
    for (auto& x : th) x.join();
    std::printf("bad = %d\n", bad);
    return bad != 0;
}

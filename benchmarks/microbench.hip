// microbench.hip -- design probes for the quantification kernel on MI355X (not part of the library).
//   stream  : coalesced 16 B/lane read of a large buffer (the col_idx / row_ptr stream)
//   gather  : random 8-byte gathers from a table of T doubles through L1/L2 (logp lookups)
//   window  : the same gathers confined to a sliding window of W entries (column-window locality)
//   lds     : random 8-byte gathers from an LDS-resident table
//   ldsrmw  : random LDS read-modify-write of doubles (running sums kept in LDS)
// Build: hipcc --offload-arch=gfx950 -O3 benchmarks/microbench.hip -o benchmarks/microbench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

__global__ __launch_bounds__(256) void k_stream(const int4* __restrict__ src, size_t n_vec, double* out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    int acc = 0;
    for (; i < n_vec; i += stride) { int4 v = src[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x7fffffff) out[0] = acc;
}

template <int MODE> // 0 plain, 1 nontemporal
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ tab, uint32_t T, int iters, uint32_t window,
                                               double* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t h = hash32(gid * 0x9E3779B9U + (uint32_t)(it * 8 + u) * 0x85EBCA6BU);
            uint32_t idx;
            if (window) {
                // the whole grid sweeps the table in lockstep-ish windows
                uint32_t base = (uint32_t)(((uint64_t)it * (T - window)) / (uint32_t)(iters > 1 ? iters - 1 : 1));
                idx = base + (uint32_t)(((uint64_t)h * window) >> 32);
            } else {
                idx = (uint32_t)(((uint64_t)h * T) >> 32);
            }
            v[u] = MODE == 1 ? __builtin_nontemporal_load(tab + idx) : tab[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_lds_gather(const double* __restrict__ tab, uint32_t T, int iters, double* out) {
    extern __shared__ __attribute__((aligned(16))) double s_tab[];
    for (uint32_t j = threadIdx.x; j < T; j += 256) s_tab[j] = tab[j];
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t h = hash32(gid * 0x9E3779B9U + (uint32_t)(it * 8 + u) * 0x85EBCA6BU);
            acc += s_tab[(uint32_t)(((uint64_t)h * T) >> 32)];
        }
    }
    if (acc == 12345.678) out[0] = acc;
}

__global__ __launch_bounds__(1024) void k_lds_rmw(uint32_t S, int iters, double* out) {
    extern __shared__ __attribute__((aligned(16))) double s_sum[];
    for (uint32_t j = threadIdx.x; j < S; j += 1024) s_sum[j] = 0.0;
    __syncthreads();
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        // conflict-free by construction within a step: slot = perm(step) + tid
        uint32_t h = hash32((uint32_t)it * 0x85EBCA6BU + blockIdx.x);
        uint32_t slot = (h + threadIdx.x * 17u) % S; // distinct per thread while 17*1024 < ... not needed for timing
        s_sum[slot] += (double)gid;
        if ((it & 3) == 3) __syncthreads();
    }
    __syncthreads();
    if (s_sum[threadIdx.x] == 12345.678) out[0] = 1.0;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double* d_out; CK(hipMalloc(&d_out, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d MHz L2 %d LDS/blk %zu\n", prop.gcnArchName, prop.multiProcessorCount,
           prop.clockRate / 1000, prop.l2CacheSize, prop.sharedMemPerBlock);
    const int grid = prop.multiProcessorCount * 8;

    { // stream
        const size_t bytes = (size_t)3200 << 20;
        int4* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes));
        for (int g : {grid, grid * 2, grid * 4}) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_stream, dim3(g), dim3(256), 0, 0, d, bytes / 16, d_out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                if (rep == 2) printf("stream  grid %5d: %.3f ms  %.1f GB/s\n", g, time_ms(e0, e1), bytes / time_ms(e0, e1) / 1e6);
            }
        }
        CK(hipFree(d));
    }
    const uint32_t Ts[] = {1000, 8192, 100000, 1000000};
    double* d_tab; CK(hipMalloc(&d_tab, 8u * 1000000)); CK(hipMemset(d_tab, 0, 8u * 1000000));
    const int iters = 36; // 2048 blocks * 256 thr * 36 * 8 = 1.5e8 gathers
    for (uint32_t T : Ts) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, d_tab, T, iters, 0u, d_out);
                else hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, d_tab, T, iters, 0u, d_out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                if (rep == 2) { double n = (double)grid * 256 * iters * 8; printf("gather  T %7u mode %d: %.3f ms  %.2f Ggather/s\n", T, mode, time_ms(e0, e1), n / time_ms(e0, e1) / 1e6); }
            }
        }
    }
    for (uint32_t W : {512u, 4096u, 16384u}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, d_tab, 100000u, iters, W, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double n = (double)grid * 256 * iters * 8; printf("window  W %7u (T 100000): %.3f ms  %.2f Ggather/s\n", W, time_ms(e0, e1), n / time_ms(e0, e1) / 1e6); }
        }
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_gather), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (uint32_t T : {1000u, 8192u, 16384u}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_lds_gather, dim3(grid), dim3(256), T * 8, 0, d_tab, T, iters, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double n = (double)grid * 256 * iters * 8; printf("lds     T %7u: %.3f ms  %.2f Ggather/s\n", T, time_ms(e0, e1), n / time_ms(e0, e1) / 1e6); }
        }
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_rmw), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (uint32_t S : {8192u, 16384u}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_lds_rmw, dim3(prop.multiProcessorCount), dim3(1024), S * 8, 0, S, 2048, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double n = (double)prop.multiProcessorCount * 1024 * 2048; printf("ldsrmw  S %7u: %.3f ms  %.2f Grmw/s\n", S, time_ms(e0, e1), n / time_ms(e0, e1) / 1e6); }
        }
    }
    return 0;
}

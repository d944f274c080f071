// microbench4.hip -- 4-byte vs 8-byte column-local gathers, and LDS integer atomic adds (design probe
// for the single-precision / fixed-point filter pass).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <typename T>
__global__ __launch_bounds__(256) void k_segperm(const T* __restrict__ tab, uint32_t Tn, uint32_t SEG, int iters, T* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = gid >> 6, lane = gid & 63;
    const uint32_t n_seg = Tn / SEG;
    T acc = 0;
    for (int it = 0; it < iters; ++it) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t step = (uint32_t)(it * 8 + u);
            const uint32_t seg = (uint32_t)(((uint64_t)step * n_seg) / (uint32_t)(iters * 8)); // all waves sweep together
            const uint32_t within = hash32(lane * 0x27d4eb2fU + step * 0x165667b1U + wave) % SEG;
            v[u] = tab[seg * SEG + within];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == (T)12345) out[0] = acc;
}

template <int KIND> // 0 u32, 1 u64, 2 f64
__global__ __launch_bounds__(1024) void k_lds_add(uint32_t S, int iters, double* out) {
    extern __shared__ __attribute__((aligned(16))) double s_sum[];
    for (uint32_t j = threadIdx.x; j < S; j += 1024) s_sum[j] = 0.0;
    __syncthreads();
    unsigned* s_u32 = reinterpret_cast<unsigned*>(s_sum);
    unsigned long long* s_u64 = reinterpret_cast<unsigned long long*>(s_sum);
    for (int it = 0; it < iters; ++it) {
        const uint32_t h = hash32((uint32_t)it * 0x85EBCA6BU + blockIdx.x * 77u + threadIdx.x * 0x9E3779B9U);
        const uint32_t slot = h % S;
        if (KIND == 0) __hip_atomic_fetch_add(&s_u32[slot], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (KIND == 1) __hip_atomic_fetch_add(&s_u64[slot], 3ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else __hip_atomic_fetch_add(&s_sum[slot], 1.5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (s_sum[threadIdx.x] == 12345.678) out[0] = 1.0;
}
static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double* d_out; CK(hipMalloc(&d_out, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount, grid = n_cu * 8;
    double* d_tab; CK(hipMalloc(&d_tab, 8u * 131072)); CK(hipMemset(d_tab, 0, 8u * 131072));
    const int iters = 36;
    for (uint32_t SEG : {64u, 256u}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_segperm<double>, dim3(grid), dim3(256), 0, 0, d_tab, 100000u, SEG, iters, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) printf("gather 8B SEG %4u: %.3f ms  %.2f Ggather/s\n", SEG, time_ms(e0, e1), (double)grid * 256 * iters * 8 / time_ms(e0, e1) / 1e6);
        }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_segperm<float>, dim3(grid), dim3(256), 0, 0, (const float*)d_tab, 100000u, SEG, iters, (float*)d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) printf("gather 4B SEG %4u: %.3f ms  %.2f Ggather/s\n", SEG, time_ms(e0, e1), (double)grid * 256 * iters * 8 / time_ms(e0, e1) / 1e6);
        }
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_add<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_add<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_add<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const char* names[3] = {"ds_add_u32", "ds_add_u64", "ds_add_f64"};
    for (int kind = 0; kind < 3; ++kind)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(k_lds_add<0>, dim3(n_cu), dim3(1024), 16384 * 8, 0, 16384u, 4096, d_out);
            if (kind == 1) hipLaunchKernelGGL(k_lds_add<1>, dim3(n_cu), dim3(1024), 16384 * 8, 0, 16384u, 4096, d_out);
            if (kind == 2) hipLaunchKernelGGL(k_lds_add<2>, dim3(n_cu), dim3(1024), 16384 * 8, 0, 16384u, 4096, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double n = (double)n_cu * 1024 * 4096; float ms = time_ms(e0, e1);
                printf("%s random: %.3f ms  %.2f Gop/s (%.1f clk per wave-instr per CU)\n", names[kind], ms, n / ms / 1e6, ms * 1e-3 * 2.4e9 / (n / 64 / n_cu)); }
        }
    return 0;
}

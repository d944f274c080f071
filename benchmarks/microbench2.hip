// microbench2.hip -- second round of design probes (not part of the library):
//   segperm  : 8-byte gathers where the 64 lanes of a wave fall in one SEG-entry segment of the table
//              (a permutation inside it); segments random (L2) or advancing (L1/L2 friendly)
//   ldsadd   : ds_add_f64 (LDS fp64 atomic, no return) / ds_add_u64 / read+add+write to random slots
//   proto    : stream 4-byte packed entries + column-sorted gather + LDS fp64 atomic add
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

template <int SEQ>
__global__ __launch_bounds__(256) void k_segperm(const double* __restrict__ tab, uint32_t T, uint32_t SEG, int iters, double* out) {
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave = gid >> 6, lane = gid & 63;
    const uint32_t n_seg = T / SEG;
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t step = (uint32_t)(it * 8 + u);
            uint32_t seg;
            if (SEQ) seg = (uint32_t)(((uint64_t)step * n_seg) / (uint32_t)(iters * 8)) ; // all waves sweep together
            else seg = hash32(wave * 0x9E3779B9U + step * 0x85EBCA6BU) % n_seg;
            const uint32_t within = hash32(lane * 0x27d4eb2fU + step * 0x165667b1U + wave) % SEG;
            v[u] = tab[seg * SEG + within];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    if (acc == 12345.678) out[0] = acc;
}

template <int KIND> // 0: ds_add_f64, 1: ds_add_u64, 2: read+add+write
__global__ __launch_bounds__(1024) void k_lds_add(uint32_t S, int iters, double* out) {
    extern __shared__ __attribute__((aligned(16))) double s_sum[];
    for (uint32_t j = threadIdx.x; j < S; j += 1024) s_sum[j] = 0.0;
    __syncthreads();
    unsigned long long* s_u = reinterpret_cast<unsigned long long*>(s_sum);
    for (int it = 0; it < iters; ++it) {
        const uint32_t h = hash32((uint32_t)it * 0x85EBCA6BU + blockIdx.x * 77u + threadIdx.x * 0x9E3779B9U);
        const uint32_t slot = h % S;
        if (KIND == 0) __hip_atomic_fetch_add(&s_sum[slot], 1.5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (KIND == 1) __hip_atomic_fetch_add(&s_u[slot], 3ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else s_sum[slot] += 1.5;
        if ((it & 7) == 7) __syncthreads();
    }
    __syncthreads();
    if (s_sum[threadIdx.x] == 12345.678) out[0] = 1.0;
}

// prototype of the windowed-slab inner loop: entries = (col << 14 | row_local), 4 per lane per step
__global__ __launch_bounds__(1024) void k_proto(const uint32_t* __restrict__ entries, size_t n_per_block,
                                                const double* __restrict__ tab, double* out, int barrier_every) {
    extern __shared__ __attribute__((aligned(16))) double s_sum[];
    for (uint32_t j = threadIdx.x; j < 16384; j += 1024) s_sum[j] = 0.0;
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(entries + (size_t)blockIdx.x * n_per_block);
    const size_t n_vec = n_per_block / 4;
    uint4 nxt = src[threadIdx.x];
    int step = 0;
    for (size_t i = threadIdx.x; i < n_vec; i += 1024, ++step) {
        const uint4 cur = nxt;
        if (i + 1024 < n_vec) nxt = src[i + 1024];
        const uint32_t e[4] = {cur.x, cur.y, cur.z, cur.w};
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = tab[e[u] >> 14];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __hip_atomic_fetch_add(&s_sum[e[u] & 16383], v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (barrier_every && (step % barrier_every) == barrier_every - 1) __syncthreads();
    }
    __syncthreads();
    double acc = 0;
    for (uint32_t j = threadIdx.x; j < 16384; j += 1024) acc += s_sum[j];
    if (acc == 12345.678) out[0] = 1.0;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double* d_out; CK(hipMalloc(&d_out, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount, grid = n_cu * 8;
    double* d_tab; CK(hipMalloc(&d_tab, 8u * 131072)); CK(hipMemset(d_tab, 0, 8u * 131072));
    const int iters = 36;
    for (uint32_t SEG : {16u, 64u, 256u, 1024u}) {
        for (int seq = 0; seq < 2; ++seq) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                if (seq) hipLaunchKernelGGL(k_segperm<1>, dim3(grid), dim3(256), 0, 0, d_tab, 100000u, SEG, iters, d_out);
                else hipLaunchKernelGGL(k_segperm<0>, dim3(grid), dim3(256), 0, 0, d_tab, 100000u, SEG, iters, d_out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                if (rep == 2) { double n = (double)grid * 256 * iters * 8; printf("segperm SEG %5u seq %d: %.3f ms  %.2f Ggather/s\n", SEG, seq, time_ms(e0, e1), n / time_ms(e0, e1) / 1e6); }
            }
        }
    }
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_add<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_add<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lds_add<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(k_lds_add<0>, dim3(n_cu), dim3(1024), 16384 * 8, 0, 16384u, 4096, d_out);
            if (kind == 1) hipLaunchKernelGGL(k_lds_add<1>, dim3(n_cu), dim3(1024), 16384 * 8, 0, 16384u, 4096, d_out);
            if (kind == 2) hipLaunchKernelGGL(k_lds_add<2>, dim3(n_cu), dim3(1024), 16384 * 8, 0, 16384u, 4096, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double n = (double)n_cu * 1024 * 4096; printf("ldsadd kind %d: %.3f ms  %.2f Gop/s\n", kind, time_ms(e0, e1), n / time_ms(e0, e1) / 1e6); }
        }
    }
    { // prototype: per block 16384 rows * 6 = 98304 entries sorted by column, several slabs per block
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_proto), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        const size_t per_block = 98304 * 24; // 24 slabs worth of entries per block
        const size_t total = per_block * n_cu;
        uint32_t* h = (uint32_t*)malloc(total * 4);
        for (size_t b = 0; b < (size_t)n_cu; ++b)
            for (size_t i = 0; i < per_block; ++i) {
                size_t in_slab = i % 98304;
                uint32_t col = (uint32_t)((in_slab * 100000ull) / 98304); // sorted by column within the slab
                uint32_t row = (uint32_t)((in_slab * 2654435761ull + b * 7919) % 16384);
                h[b * per_block + i] = (col << 14) | row;
            }
        uint32_t* d; CK(hipMalloc(&d, total * 4)); CK(hipMemcpy(d, h, total * 4, hipMemcpyHostToDevice));
        for (int be : {0, 1, 2, 4}) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_proto, dim3(n_cu), dim3(1024), 16384 * 8, 0, d, per_block, d_tab, d_out, be);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                if (rep == 2) printf("proto barrier_every %d: %.3f ms  %.2f Gentry/s  stream %.1f GB/s (canonical-equivalent %.1f GB/s)\n", be,
                                     time_ms(e0, e1), total / time_ms(e0, e1) / 1e6, total * 4.0 / time_ms(e0, e1) / 1e6,
                                     total / 6.0 * 32.0 / time_ms(e0, e1) / 1e6);
            }
        }
    }
    return 0;
}

// microbench3.hip -- LDS fp64 atomic add: random rows vs bank-conflict-free rows (design probe).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

// MODE 4: ds_add_f32 random slot; MODE 0: random slot; 1: conflict-free (slot mod 32 == lane mod 32, random multiple of 32 on top);
// 2: like 1 but per 64-lane: slot mod 64 == lane; 3: random slot but read+add+write
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_add(uint32_t S, int iters, double* out) {
    extern __shared__ __attribute__((aligned(16))) double s_sum[];
    for (uint32_t j = threadIdx.x; j < S; j += WAVES * 64) s_sum[j] = 0.0;
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
        const uint32_t h = hash32((uint32_t)it * 0x85EBCA6BU + blockIdx.x * 77u + threadIdx.x * 0x9E3779B9U);
        uint32_t slot;
        if (MODE == 0 || MODE == 3 || MODE == 4) slot = h % S;
        else if (MODE == 1) slot = ((h % (S / 32)) * 32) + (lane & 31);
        else slot = ((h % (S / 64)) * 64) + lane;
        if (MODE == 4) __hip_atomic_fetch_add(reinterpret_cast<float*>(s_sum) + slot, 1.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        else if (MODE == 3) s_sum[slot] += 1.5;
        else __hip_atomic_fetch_add(&s_sum[slot], 1.5, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
    __syncthreads();
    if (s_sum[threadIdx.x] == 12345.678) out[0] = 1.0;
}
static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
template <int MODE, int WAVES> void run(const char* name, int n_cu, double* d_out, hipEvent_t e0, hipEvent_t e1) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_add<MODE, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int iters = 4096 * 16 / WAVES;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_add<MODE, WAVES>), dim3(n_cu), dim3(WAVES * 64), 16384 * 8, 0, 16384u, iters, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        if (rep == 2) { double n = (double)n_cu * WAVES * 64 * iters; float ms = time_ms(e0, e1);
            printf("%-28s waves %2d: %.3f ms  %.2f Gop/s  (%.1f clk per wave-instr per CU @2.4GHz)\n", name, WAVES, ms, n / ms / 1e6, ms * 1e-3 * 2.4e9 / (n / 64 / n_cu)); }
    }
}
int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double* d_out; CK(hipMalloc(&d_out, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    run<0, 16>("ds_add_f64 random", n_cu, d_out, e0, e1);
    run<1, 16>("ds_add_f64 conflict-free/32", n_cu, d_out, e0, e1);
    run<2, 16>("ds_add_f64 conflict-free/64", n_cu, d_out, e0, e1);
    run<3, 16>("rd+add+wr random", n_cu, d_out, e0, e1);
    run<4, 16>("ds_add_f32 random", n_cu, d_out, e0, e1);
    run<0, 4>("ds_add_f64 random", n_cu, d_out, e0, e1);
    run<1, 4>("ds_add_f64 conflict-free/32", n_cu, d_out, e0, e1);
    run<2, 4>("ds_add_f64 conflict-free/64", n_cu, d_out, e0, e1);
    run<3, 4>("rd+add+wr random", n_cu, d_out, e0, e1);
    return 0;
}

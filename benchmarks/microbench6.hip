// microbench6.hip -- HBM streaming rate of the slab passes' access pattern as a function of the bytes per lane of one
// load (16 / 12 / 8) and of the chunks per wave in flight: 256 workgroups x 16 waves, workgroup b streams its own
// contiguous region, wave w takes chunks w, w + 16, ...; U chunks per step, the next step's loads issued before the
// current step is consumed (the structure of k_fr_filter_q8).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int DW> struct Vec { uint32_t v[DW]; };

template <int DW, int U, int DEPTH> // DEPTH = steps in flight ahead of the one being consumed (1 or 2)
__global__ __launch_bounds__(1024) void k_stream(const uint32_t* __restrict__ buf, int64_t chunks_per_wg, uint32_t* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t* base = buf + (int64_t)blockIdx.x * chunks_per_wg * 64 * DW;
    uint32_t acc = 0;
    Vec<DW> q[DEPTH + 1][U];
    auto load = [&](Vec<DW>(&dst)[U], int64_t cb) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int64_t c = cb + (int64_t)u * 16;
            c = c < chunks_per_wg ? c : 0;
            const uint32_t* p = base + (c * 64 + lane) * DW;
#pragma unroll
            for (int d = 0; d < DW; ++d) dst[u].v[d] = p[d];
        }
    };
    int64_t cb = wave;
    load(q[0], cb);
    if (DEPTH == 2) load(q[1], cb + 16 * U);
    for (; cb < chunks_per_wg; cb += 16 * U) {
        load(q[DEPTH], cb + (int64_t)DEPTH * 16 * U);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int d = 0; d < DW; ++d) acc ^= q[0][u].v[d];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s)
#pragma unroll
            for (int u = 0; u < U; ++u) q[s][u] = q[s + 1][u];
    }
    if (acc == 0x12345u) out[0] = acc;
}
static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
template <int DW, int U, int DEPTH>
void run(const uint32_t* d_buf, size_t bytes, uint32_t* d_out, hipEvent_t e0, hipEvent_t e1) {
    const int64_t chunks_per_wg = (int64_t)(bytes / 256 / (64 * DW * 4));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<DW, U, DEPTH>), dim3(256), dim3(1024), 0, 0, d_buf, chunks_per_wg, d_out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        const float ms = time_ms(e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double moved = (double)chunks_per_wg * 256 * 64 * DW * 4;
    printf("%2d B per lane, U %d, depth %d: %.3f ms  %.2f TB/s  (%.0f KB in flight per CU)\n", DW * 4, U, DEPTH, best, moved / best / 1e9,
           16.0 * U * DEPTH * 64 * DW * 4 / 1024);
}
int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    const size_t bytes = (size_t)2400 << 20; // ~ the slab stream of config 3
    uint32_t* d_buf; CK(hipMalloc(&d_buf, bytes)); CK(hipMemset(d_buf, 1, bytes));
    run<4, 4, 1>(d_buf, bytes, d_out, e0, e1);
    run<4, 2, 1>(d_buf, bytes, d_out, e0, e1);
    run<4, 8, 1>(d_buf, bytes, d_out, e0, e1);
    run<4, 4, 2>(d_buf, bytes, d_out, e0, e1);
    run<3, 4, 1>(d_buf, bytes, d_out, e0, e1);
    run<3, 6, 1>(d_buf, bytes, d_out, e0, e1);
    run<3, 4, 2>(d_buf, bytes, d_out, e0, e1);
    run<2, 4, 1>(d_buf, bytes, d_out, e0, e1);
    run<2, 8, 1>(d_buf, bytes, d_out, e0, e1);
    run<2, 8, 2>(d_buf, bytes, d_out, e0, e1);
    run<1, 8, 2>(d_buf, bytes, d_out, e0, e1);
    return 0;
}

// microbench5.hip -- how fast does the L2 feed a table that is far larger than a CU's L1? (design probe for the
// FIXED_SUM pass: every 8192-row slab walks the whole 0.8 MB fixed-point table once, in column order.)
//   mode 0: every workgroup reads the table front to back, 16 B per lane, fully coalesced
//   mode 1: 8 B per lane, one wave instruction covers LINES consecutive 128-byte lines, lanes spread at random in them
//           (the access pattern of the column-sorted chunk stream: LINES = 8 at E = 1e5, 4 at E = 5e4, ...)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

__global__ __launch_bounds__(1024) void k_sweep16(const uint4* __restrict__ tab, uint32_t n16, int sweeps, uint32_t* out) {
    uint32_t acc = 0;
    const uint32_t start = (hash32(blockIdx.x) % n16) & ~1023u;
    for (int s = 0; s < sweeps; ++s)
        for (uint32_t i = threadIdx.x; i < n16; i += 4096) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { uint32_t j = start + i + u * 1024; j = j >= n16 ? j - n16 : j; v[u] = tab[j]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    if (acc == 12345u) out[0] = acc;
}

template <int LINES, int MODE> // MODE 0: lanes at random in the LINES lines; 1: lanes in ascending address order (sorted columns); 2: sorted, 4-byte entries
__global__ __launch_bounds__(1024) void k_sweep_gather(const unsigned long long* __restrict__ tab, uint32_t n_lines, int sweeps, uint32_t* out) {
    unsigned long long acc = 0;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t start = hash32(blockIdx.x) % n_lines;
    const uint32_t groups = n_lines / LINES; // one instruction per group of LINES lines, waves take groups round robin
    for (int s = 0; s < sweeps; ++s)
        for (uint32_t g = wave; g < groups; g += 16 * 4) {
            unsigned long long v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t gg = g + u * 16; gg = gg >= groups ? gg - groups : gg;
                uint32_t line = start + gg * LINES + (hash32(lane * 77u + gg) % LINES); line = line >= n_lines ? line - n_lines : line;
                if (MODE == 0) v[u] = tab[(size_t)line * 16 + (hash32(lane + gg * 131u) & 15u)];
                else {
                    uint32_t l0 = start + gg * LINES; l0 = l0 >= n_lines ? l0 - n_lines : l0; // (a window may run past the end by < LINES lines: the table is padded)
                    constexpr uint32_t SPAN = LINES * 16 / 64; // slots per lane
                    const uint32_t slot = lane * SPAN + (SPAN > 1 ? hash32(lane + gg * 131u) % (SPAN > 1 ? SPAN : 1) : 0u);
                    if (MODE == 1) v[u] = tab[(size_t)l0 * 16 + slot];
                    else v[u] = reinterpret_cast<const uint32_t*>(tab)[(size_t)l0 * 32 + slot * 2 + (hash32(lane ^ gg) & 1u)];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    if (acc == 12345ULL) out[0] = (uint32_t)acc;
}
static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }
int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    uint32_t* d_out; CK(hipMalloc(&d_out, 64));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    for (uint32_t E : {12500u, 50000u, 100000u, 200000u}) {
        const size_t bytes = (size_t)E * 8;
        void* d_tab; CK(hipMalloc(&d_tab, bytes + 4096)); CK(hipMemset(d_tab, 1, bytes + 4096));
        const int sweeps = 48;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_sweep16, dim3(n_cu), dim3(1024), 0, 0, (const uint4*)d_tab, (uint32_t)(bytes / 16), sweeps, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) printf("E %6u coalesced 16B: %.3f ms  %.2f TB/s into the CUs\n", E, time_ms(e0, e1), (double)n_cu * sweeps * bytes / time_ms(e0, e1) / 1e9);
        }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_sweep_gather<8, 0>), dim3(n_cu), dim3(1024), 0, 0, (const unsigned long long*)d_tab, (uint32_t)(bytes / 128), sweeps, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double ms = time_ms(e0, e1), instr = (double)n_cu * sweeps * (bytes / 128 / 8);
                printf("E %6u gather 8 lines/instr: %.3f ms  %.2f TB/s of lines, %.1f clk per wave-instr per CU, %.2f Tgather/s\n", E, ms, (double)n_cu * sweeps * bytes / ms / 1e9,
                       ms * 1e-3 * 2.4e9 / (instr / n_cu), instr * 64 / ms / 1e9); }
        }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_sweep_gather<4, 0>), dim3(n_cu), dim3(1024), 0, 0, (const unsigned long long*)d_tab, (uint32_t)(bytes / 128), sweeps, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double ms = time_ms(e0, e1), instr = (double)n_cu * sweeps * (bytes / 128 / 4);
                printf("E %6u gather 4 lines/instr: %.3f ms  %.2f TB/s of lines, %.1f clk per wave-instr per CU, %.2f Tgather/s\n", E, ms, (double)n_cu * sweeps * bytes / ms / 1e9,
                       ms * 1e-3 * 2.4e9 / (instr / n_cu), instr * 64 / ms / 1e9); }
        }
        for (int mode = 1; mode <= 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 1) hipLaunchKernelGGL((k_sweep_gather<8, 1>), dim3(n_cu), dim3(1024), 0, 0, (const unsigned long long*)d_tab, (uint32_t)(bytes / 128), sweeps, d_out);
            else hipLaunchKernelGGL((k_sweep_gather<8, 2>), dim3(n_cu), dim3(1024), 0, 0, (const unsigned long long*)d_tab, (uint32_t)(bytes / 128), sweeps, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double ms = time_ms(e0, e1), instr = (double)n_cu * sweeps * (bytes / 128 / 8);
                printf("E %6u SORTED gather 8 lines/instr (%s): %.3f ms  %.1f clk per wave-instr per CU, %.2f Tgather/s\n", E, mode == 1 ? "8 B" : "4 B", ms,
                       ms * 1e-3 * 2.4e9 / (instr / n_cu), instr * 64 / ms / 1e9); }
        }
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_sweep_gather<4, 1>), dim3(n_cu), dim3(1024), 0, 0, (const unsigned long long*)d_tab, (uint32_t)(bytes / 128), sweeps, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            if (rep == 2) { double ms = time_ms(e0, e1), instr = (double)n_cu * sweeps * (bytes / 128 / 4);
                printf("E %6u SORTED gather 4 lines/instr (8 B, every slot): %.3f ms  %.1f clk per wave-instr per CU, %.2f Tgather/s\n", E, ms,
                       ms * 1e-3 * 2.4e9 / (instr / n_cu), instr * 64 / ms / 1e9); }
        }
        CK(hipFree(d_tab));
    }
    return 0;
}

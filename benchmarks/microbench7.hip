// microbench7.hip -- the table-gather floor of the exact all-rows passes, measured on the access pattern they issue.
//
// k_fx2 (FIXED_SUM, cut_off = 0) and every exact fp64 pass gather ONE 8-byte table entry per CSR entry through the
// texture path: 6e8 lanes = 9.4e6 64-lane gather instructions per pass at config 3. What one such instruction costs a CU
// depends on how many 128-byte lines its 64 addresses touch, i.e. on how many column-sorted entries the slab holds:
//   S entries sorted by column over E columns -> 64 consecutive entries span 64 * E / S columns.
// This benchmark builds exactly that stream -- slabs of S packed entries `col << 14 | slot`, column-sorted inside the slab,
// cut into 1 KiB chunks transposed like the product's (lane k's 16 bytes = sorted positions k, 64+k, 128+k, 192+k) -- and
// runs the adding half of k_fx2 on it: 12 waves of a 16-wave workgroup stream chunks (16 B per lane, U chunks in flight)
// and issue four 8-byte gathers per chunk, nothing else (no LDS adds, no fold). Columns are stratified-uniform
// (col_i = floor(E * (i + u_i) / S)): the density of a sorted uniform sample without its gap variance.
//   S = 49152   the 8192-row slabs of k_fx2 / k_fr_filter                        (8 lines per instruction at E = 1e5)
//   S = 98304   16384-row slabs (the gathered fp64 filter)
//   S = 16384   the value-scatter slabs (k_vs: every gathered value needs an 8-byte LDS slot)
//   S = 8704    what a DOUBLE-BUFFERED value scatter could hold in 160 KB of LDS
//   dense       64 consecutive columns per instruction: the texture path's best case
// Output: ns and clocks per gather instruction per CU, and the time 9.4e6 instructions take on 256 CUs.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __host__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}

// stream[slab][chunk][lane][4]: sorted position q = chunk * 256 + u * 64 + lane of the slab
__global__ void k_build(uint32_t* stream, int64_t n_slabs, int S, uint32_t E, int dense, int w12 = 0) {
    const int64_t n = n_slabs * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slab = i / S;
        const int q = (int)(i - slab * S);
        uint32_t col;
        if (dense) col = (uint32_t)((mix64((uint64_t)slab * 977u + q / 64) % (E - 64)) + (q & 63));
        else {
            const double u = (double)(mix64((uint64_t)i) >> 11) * 0x1.0p-53;
            col = (uint32_t)(((double)q + u) * (double)E / (double)S);
            if (col >= E) col = E - 1;
        }
        const int chunk = q >> 8, r = q & 255, uu = r >> 6, lane = r & 63;
        if (w12) { if (uu < 3) stream[(slab * (S / 256) + chunk) * 192 + lane * 3 + uu] = (col << 14) | (uint32_t)(q & 16383); } // 12 B per lane
        else stream[(slab * (S / 256) + chunk) * 256 + lane * 4 + uu] = (col << 14) | (uint32_t)(q & 16383);
    }
}

template <int U>
__global__ __launch_bounds__(1024) void k_gather(const uint4* __restrict__ stream, const unsigned long long* __restrict__ tab,
                                                 int64_t n_slabs, int chunks_per_slab, int64_t resident_slabs, unsigned long long* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 12) return; // the four folding waves of k_fx2 are absent here
    unsigned long long acc = 0;
    for (int64_t s = blockIdx.x; s < n_slabs; s += gridDim.x) {
        // resident_slabs > 0: the same few slabs over and over (their stream stays in the L2 / Infinity Cache): the gathers alone
        const uint4* base = stream + (resident_slabs > 0 ? s % resident_slabs : s) * chunks_per_slab * 64;
        for (int c0 = wave * U; c0 < chunks_per_slab; c0 += 12 * U) {
            uint4 q[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u < chunks_per_slab ? c0 + u : chunks_per_slab - 1;
                q[u] = base[(size_t)c * 64 + lane];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const unsigned long long a = tab[q[u].x >> 14], b = tab[q[u].y >> 14], c = tab[q[u].z >> 14], d = tab[q[u].w >> 14];
                acc += (a ^ b) + (c ^ d);
            }
        }
    }
    if (acc == 0x12345ULL) out[0] = acc;
}

// the same with 12 bytes per lane and chunk (what a 24-bit form of the stream would cost k_fx2): three dwords loaded, four gathers
// (the fourth column next to the first: the lines an instruction touches stay what they are)
template <int U>
__global__ __launch_bounds__(1024) void k_gather24(const uint32_t* __restrict__ stream, const unsigned long long* __restrict__ tab,
                                                   int64_t n_slabs, int chunks_per_slab, int64_t resident_slabs, unsigned long long* out, uint32_t E) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave >= 12) return;
    unsigned long long acc = 0;
    for (int64_t s = blockIdx.x; s < n_slabs; s += gridDim.x) {
        const uint32_t* base = stream + (resident_slabs > 0 ? s % resident_slabs : s) * chunks_per_slab * 192;
        for (int c0 = wave * U; c0 < chunks_per_slab; c0 += 12 * U) {
            uint32_t q[U][3];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u < chunks_per_slab ? c0 + u : chunks_per_slab - 1;
                const uint32_t* p = base + ((size_t)c * 64 + lane) * 3;
                q[u][0] = p[0]; q[u][1] = p[1]; q[u][2] = p[2];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t c4 = (q[u][2] >> 14) + 130u < E ? (q[u][2] >> 14) + 130u : 0u; // (where sorted position 192 + lane would be)
                const unsigned long long a = tab[q[u][0] >> 14], b = tab[q[u][1] >> 14], c = tab[q[u][2] >> 14], d = tab[c4];
                acc += (a ^ b) + (c ^ d);
            }
        }
    }
    if (acc == 0x12345ULL) out[0] = acc;
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const double clk_ghz = prop.clockRate * 1e-6;
    const uint32_t E = 100000;
    const int64_t total = 600000000LL / 256 * 256;
    uint32_t* d_stream; CK(hipMalloc(&d_stream, (size_t)total * 4 + (1 << 22)));
    unsigned long long* d_tab; CK(hipMalloc(&d_tab, (size_t)E * 8));
    CK(hipMemset(d_tab, 1, (size_t)E * 8));
    unsigned long long* d_out; CK(hipMalloc(&d_out, 64));
    printf("# %s, %d CUs, %.2f GHz; table E = %u x 8 B; %lld entries per pass (config 3: 6e8); 12 gathering waves per CU, 4 chunks in flight\n",
           prop.name, n_cu, clk_ghz, E, (long long)total);
    struct Case { int S; int dense; const char* what; };
    const Case cases[] = {{49152, 0, "8192-row slabs (k_fx2, k_fr_filter)"}, {98304, 0, "16384-row slabs (gathered fp64 filter)"},
                          {16384, 0, "value-scatter slabs (k_vs)"}, {8704, 0, "double-buffered value-scatter slabs"},
                          {4096, 0, "a wave-slab sub-stream of 4096 entries"}, {49152, 1, "DENSE: 64 consecutive columns per instruction"}};
    for (const Case& cs : cases) {
        const int64_t n_slabs = total / cs.S;
        const int cps = cs.S / 256;
        hipLaunchKernelGGL(k_build, dim3(n_cu * 8), dim3(1024), 0, 0, d_stream, n_slabs, cs.S, E, cs.dense);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        for (int resident = 0; resident < 2; ++resident) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_gather<4>, dim3(n_cu), dim3(1024), 0, 0, reinterpret_cast<const uint4*>(d_stream), d_tab, n_slabs, cps,
                                   resident ? (int64_t)n_cu : (int64_t)0, d_out);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            if (cs.S == 49152 && !cs.dense) { // the 12-byte form of the same stream: 3/4 of the bytes, the same gathers
                hipLaunchKernelGGL(k_build, dim3(n_cu * 8), dim3(1024), 0, 0, d_stream, n_slabs, cs.S, E, cs.dense, 1);
                CK(hipGetLastError());
                float b24 = 1e9f;
                for (int rep = 0; rep < 6; ++rep) {
                    CK(hipEventRecord(e0));
                    hipLaunchKernelGGL(k_gather24<4>, dim3(n_cu), dim3(1024), 0, 0, (const uint32_t*)d_stream, d_tab, n_slabs, cps,
                                       resident ? (int64_t)n_cu : (int64_t)0, d_out, E);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < b24) b24 = ms;
                }
                printf("S = %6d  the same with 12 B per lane and chunk (1.8 GB)        %-26s %.3f ms\n", cs.S, resident ? "[stream cache-resident]" : "[stream from HBM]", b24);
                hipLaunchKernelGGL(k_build, dim3(n_cu * 8), dim3(1024), 0, 0, d_stream, n_slabs, cs.S, E, cs.dense, 0);
                CK(hipGetLastError());
                CK(hipDeviceSynchronize());
            }
            const double instr = (double)n_slabs * cs.S / 64.0;            // 64-lane gather instructions
            const double per_cu_ns = best * 1e6 / (instr / n_cu);
            const double lines = 64.0 * E / cs.S / 16.0 + 1.0;
            printf("S = %6d  %-48s %-26s %.3f ms for %.3g gather instr: %.1f ns = %.1f clk per instr per CU (~%.0f lines / instr); 9.4e6 instr on 256 CUs: %.3f ms\n",
                   cs.S, cs.what, resident ? "[stream cache-resident]" : "[stream from HBM, 2.4 GB]", best, instr, per_cu_ns, per_cu_ns * clk_ghz,
                   cs.dense ? 4.0 : lines, 9.375e6 / 256.0 * per_cu_ns * 1e-6);
        }
    }
    return 0;
}

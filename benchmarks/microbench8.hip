// microbench8.hip -- can a WORKGROUP-wide slab keep fp64 row sums in CSR order? The band-synchronised form, bare.
//
// K1 as literally stated needs s_i = the sequential fp64 sum of a row's logp values in CSR (= ascending column) order. The
// one kernel that delivers it near its bounds (k_ws) keeps a wave's 4096 rows private so that the wave's in-order LDS adds
// ARE the CSR order -- which fixes the density of its table gathers at 24.5 K entries per sorted stream (~17 lines and ~45
// clocks per gather instruction, benchmarks/microbench7.hip). A workgroup-wide slab of 16384 rows (98 K entries: ~5 lines,
// ~24 clocks) needs another way to keep the order: cut the slab's column-sorted stream into BANDS of 1024 entries in which
// every row appears at most once (an entry whose predecessor sits in the same band is deferred to the next one), 16 waves
// take one 64-entry group each, and a workgroup barrier separates the bands. This benchmark runs the bare loop of that
// design on a synthetic stream of the right shape: 16 waves, stream loads one superstep (4 bands) ahead, the gathers of a
// superstep issued one superstep before their adds, per band 64 ds_add_f64 per wave (rows random: distinct with high
// probability) and `s_waitcnt lgkmcnt(0); s_barrier`. No epilogue, no row output.
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/bin/microbench8 benchmarks/microbench8.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __host__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
constexpr int kRows = 16384, kBand = 1024, kWaves = 16;

// stream[slab][superstep][wave][lane][4]: component j = the entry of band 4 * superstep + j, group `wave`, lane `lane`;
// sorted position inside the slab = band * 1024 + wave * 64 + lane
__global__ void k_build(uint32_t* stream, int64_t n_slabs, int bands, uint32_t E) {
    const int S = bands * kBand;
    const int64_t n = n_slabs * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slab = i / S;
        const int q = (int)(i - slab * S);
        const double u = (double)(mix64((uint64_t)i) >> 11) * 0x1.0p-53;
        uint32_t col = (uint32_t)(((double)q + u) * (double)E / (double)S);
        if (col >= E) col = E - 1;
        const uint32_t row = (uint32_t)(mix64((uint64_t)i * 77u + 5u) % kRows);
        const int band = q / kBand, w = (q % kBand) / 64, lane = q & 63;
        stream[((slab * (bands / 4) + band / 4) * kWaves + w) * 256 + lane * 4 + (band & 3)] = (col << 14) | row;
    }
}

template <bool BARRIER, bool ADDS, bool GATHERS>
__global__ __launch_bounds__(1024) void k_bands(const uint4* __restrict__ stream, const double* __restrict__ tab, int64_t n_slabs,
                                                int bands, double* out) {
    extern __shared__ double sums[]; // [16384]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int i = tid; i < kRows; i += 1024) sums[i] = 0.0;
    __syncthreads();
    const int ss = bands / 4; // supersteps per slab
    // the workgroup's supersteps, slab after slab: a flat sequence
    const int64_t my_slabs = (n_slabs - blockIdx.x + gridDim.x - 1) / gridDim.x;
    const int64_t total = my_slabs * ss;
    auto load = [&](int64_t k) -> uint4 {
        const int64_t kk = k < total ? k : total - 1;
        const int64_t slab = blockIdx.x + (kk / ss) * gridDim.x;
        return stream[((slab * ss + kk % ss) * kWaves + wave) * 64 + lane];
    };
    uint4 e_next = load(0);
    uint4 e_cur = e_next;
    double v[4] = {0, 0, 0, 0};
    // prologue: entries of superstep 0 -> gathers; entries of superstep 1 requested
    if (GATHERS) { v[0] = tab[e_cur.x >> 14]; v[1] = tab[e_cur.y >> 14]; v[2] = tab[e_cur.z >> 14]; v[3] = tab[e_cur.w >> 14]; }
    e_next = load(1);
    double acc = 0.0;
    for (int64_t k = 0; k < total; ++k) {
        const uint4 e_add = e_cur;       // rows of the values in v
        const double a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        e_cur = e_next;                  // superstep k + 1: its gathers go out now, its adds happen next iteration
        if (GATHERS) { v[0] = tab[e_cur.x >> 14]; v[1] = tab[e_cur.y >> 14]; v[2] = tab[e_cur.z >> 14]; v[3] = tab[e_cur.w >> 14]; }
        e_next = load(k + 2);
        const uint32_t r[4] = {e_add.x & 16383u, e_add.y & 16383u, e_add.z & 16383u, e_add.w & 16383u};
        const double a[4] = {a0, a1, a2, a3};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ADDS) __hip_atomic_fetch_add(&sums[r[j]], a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else acc += a[j];
            if (BARRIER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    }
    __syncthreads();
    double t = acc;
    for (int i = tid; i < kRows; i += 1024) t += sums[i];
    if (t == 1.2345) out[0] = t;
}

int main() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const uint32_t E = 100000;
    const int bands = 96; // 98304 entries per slab = 16384 rows x 6
    const int64_t n_slabs = 600000000LL / (bands * kBand);
    const int64_t total = n_slabs * bands * kBand;
    uint32_t* d_stream; CK(hipMalloc(&d_stream, (size_t)total * 4 + (1 << 22)));
    double* d_tab; CK(hipMalloc(&d_tab, (size_t)E * 8));
    CK(hipMemset(d_tab, 0, (size_t)E * 8));
    double* d_out; CK(hipMalloc(&d_out, 64));
    hipLaunchKernelGGL(k_build, dim3(n_cu * 8), dim3(1024), 0, 0, d_stream, n_slabs, bands, E);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    printf("# %s, %d CUs; %lld slabs of 16384 rows, %d bands of 1024 entries (%.3g entries, %.2f GB of stream); table %u x 8 B\n", prop.name, n_cu,
           (long long)n_slabs, bands, (double)total, total * 4e-9, E);
    const size_t lds = (size_t)kRows * 8;
    auto run = [&](const char* what, auto kern) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(n_cu), dim3(1024), lds, 0, reinterpret_cast<const uint4*>(d_stream), d_tab, n_slabs, bands, d_out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        CK(hipGetLastError());
        printf("%-72s %.3f ms\n", what, best);
    };
    run("stream + gathers + ds_add_f64 + barrier per band (the design)", k_bands<true, true, true>);
    run("stream + gathers + ds_add_f64, NO barriers (order lost: the bound)", k_bands<false, true, true>);
    run("stream + gathers, no adds, barrier per band", k_bands<true, false, true>);
    run("stream + gathers only", k_bands<false, false, true>);
    run("stream + ds_add_f64 + barrier per band, no gathers", k_bands<true, true, false>);
    run("stream only", k_bands<false, false, false>);
    return 0;
}

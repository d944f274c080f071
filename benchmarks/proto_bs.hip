// proto_bs.hip -- prototype of the BAND STREAM: exact CSR-order fp64 row sums with WORKGROUP-wide slabs (16384 rows) --
// layout builder + kernel + bit-exact self-check against a lane-per-row kernel, on a synthetic CSR of config 3's shape.
// (What benchmarks/microbench8.hip measured bare; standalone so that the design can be judged before it is wired in.)
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/bin/proto_bs benchmarks/proto_bs.hip && ./benchmarks/bin/proto_bs [rows] [events]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __host__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
#ifndef GPB
#define GPB 1 // 64-entry groups per wave and band: a band holds 1024 * GPB entries, a barrier follows every GPB adds
#endif
#ifndef FILL
#define FILL 928
#endif
constexpr int kSlabRows = 16384, kBand = 1024 * GPB, kWaves = 16, kRowBits = 15, kFill = FILL * GPB, kMaxBands = 4096;
constexpr uint32_t kRowMask = (1u << kRowBits) - 1;

__global__ void k_fill_cols(const int64_t* __restrict__ row_ptr, int32_t* __restrict__ col, int64_t n_rows, uint32_t E) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const int64_t b = row_ptr[r];
    const int len = (int)(row_ptr[r + 1] - b);
    uint32_t c[16];
    uint64_t ctr = (uint64_t)r * 1315423911ULL;
    for (int k = 0; k < len; ++k) {
        for (;;) {
            const uint32_t v = (uint32_t)(mix64(ctr++) % E);
            bool dup = false;
            for (int j = 0; j < k; ++j) dup |= c[j] == v;
            if (!dup) { c[k] = v; break; }
        }
    }
    for (int i = 1; i < len; ++i) { const uint32_t v = c[i]; int j = i - 1; while (j >= 0 && c[j] > v) { c[j + 1] = c[j]; --j; } c[j + 1] = v; }
    for (int k = 0; k < len; ++k) col[b + k] = (int32_t)c[k];
}

__global__ void k_ref(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col, const double* __restrict__ logp, int64_t n_rows,
                      double* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    double s = 0.0;
    for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) s += logp[col[k]];
    out[r] = s;
}

// ---- builder ------------------------------------------------------------------------------------------------------
// bands allotted to a slab (in supersteps of 4 bands)
__global__ void k_alloc(const int64_t* __restrict__ row_ptr, int64_t n_rows, int64_t n_slabs, int max_row_len, int64_t* __restrict__ ss) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slabs) return;
    const int64_t r0 = s * kSlabRows, r1 = r0 + kSlabRows < n_rows ? r0 + kSlabRows : n_rows;
    const int64_t nnz = row_ptr[r1] - row_ptr[r0];
    int64_t nom = (nnz + kFill - 1) / kFill;
    if (nom < 1) nom = 1;
    const int64_t alloc = nom + nom / 8 + max_row_len + 4;
    ss[s] = (alloc * GPB + 3) / 4;
}

// phase 1: every entry gets a band: its column's share of the slab's nominal band count, pushed later while the row's
// previous entry sits in the same or a later band, or the band is full. One workgroup per slab, counters in LDS.
__global__ __launch_bounds__(256) void k_bs_place(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col, int64_t n_rows,
                                                  uint32_t E, const int64_t* __restrict__ ss_off, uint32_t* __restrict__ tmp,
                                                  uint16_t* __restrict__ band_cnt, int* __restrict__ overflow) {
    __shared__ unsigned cnt[kMaxBands];
    const int64_t s = blockIdx.x;
    const int64_t r0 = s * kSlabRows, r1 = r0 + kSlabRows < n_rows ? r0 + kSlabRows : n_rows;
    const int n_alloc = (int)((ss_off[s + 1] - ss_off[s]) * 4 / GPB);
    const int64_t nnz = row_ptr[r1] - row_ptr[r0];
    uint64_t nom = (uint64_t)((nnz + kFill - 1) / kFill);
    if (nom < 1) nom = 1;
    for (int i = threadIdx.x; i < n_alloc; i += 256) cnt[i] = 0u;
    __syncthreads();
    uint32_t* out = tmp + ss_off[s] * 4 * 1024;
    bool ovf = false;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        int b_prev = -1;
        for (int64_t k = row_ptr[r]; k < row_ptr[r + 1] && !ovf; ++k) {
            const uint32_t c = (uint32_t)col[k];
            int b = (int)(((uint64_t)c * nom) / E);
            if (b <= b_prev) b = b_prev + 1;
            for (;;) {
                if (b >= n_alloc) { ovf = true; break; }
                const unsigned pos = atomicAdd(&cnt[b], 1u);
                if (pos < (unsigned)kBand) { out[(int64_t)b * kBand + pos] = (c << kRowBits) | (uint32_t)(r - r0); break; }
                ++b;
            }
            b_prev = b;
        }
    }
    if (ovf) atomicExch(overflow, 1);
    __syncthreads();
    for (int i = threadIdx.x; i < n_alloc; i += 256) band_cnt[ss_off[s] * 4 / GPB + i] = (uint16_t)(cnt[i] < (unsigned)kBand ? cnt[i] : kBand);
}

// phase 2: every band sorted by column (its 64-entry groups then cover consecutive columns), written transposed:
// stream[superstep][group g][lane][band & 3]. One workgroup (256 threads) per band: bitonic sort of 1024 keys in LDS.
__global__ __launch_bounds__(256) void k_bs_sort(const uint32_t* __restrict__ tmp, const uint16_t* __restrict__ band_cnt, int64_t n_bands,
                                                 uint32_t* __restrict__ stream) {
    __shared__ uint32_t key[kBand];
    const int64_t b = blockIdx.x;
    if (b >= n_bands) return;
    const int n = band_cnt[b];
    for (int i = threadIdx.x; i < kBand; i += 256) key[i] = i < n ? tmp[b * kBand + i] : 0xFFFFFFFFu;
    __syncthreads();
    for (int k = 2; k <= kBand; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < kBand / 2; t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // the lower index of pair t
                const int p = i | j;
                const bool up = (i & k) == 0;
                const uint32_t a = key[i], c = key[p];
                if ((a > c) == up) { key[i] = c; key[p] = a; }
            }
            __syncthreads();
        }
    const uint32_t last_col = n > 0 ? (key[n - 1] >> kRowBits) : 0u;
    for (int q = threadIdx.x; q < kBand; q += 256) {
        const int g = q >> 6, lane = q & 63;
        const uint32_t e = q < n ? key[q] : ((last_col << kRowBits) | (uint32_t)(kSlabRows + lane)); // padding: a sink row
        const int w = g / GPB, sub = g % GPB;
        const int64_t t = b * GPB + sub; // slot of the flat superstep sequence
        stream[(((t >> 2) * kWaves + w) * 64 + lane) * 4 + (t & 3)] = e;
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------
template <bool ATOMIC>
__global__ __launch_bounds__(1024) void k_bs(const uint4* __restrict__ stream, const int64_t* __restrict__ ss_off, const int32_t* __restrict__ ss_used,
                                             int64_t total_ss, const double* __restrict__ logp, int64_t n_rows, int64_t n_slabs, double log_cut,
                                             double* __restrict__ row_out, unsigned long long* __restrict__ kept_out) {
    extern __shared__ double sums[]; // [16384 + 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < kSlabRows + 64; i += 1024) sums[i] = 0.0;
    __syncthreads();
    const int64_t G = gridDim.x;
    // load cursor: runs two supersteps ahead of the adds, across slab boundaries; past the end: the no-op superstep
    int64_t ls = blockIdx.x, lp = 0, lp_end = 0;
    if (ls < n_slabs) { lp = ss_off[ls]; lp_end = lp + ss_used[ls]; }
    auto next_chunk = [&]() -> uint4 {
        while (lp == lp_end) {
            ls += G;
            if (ls < n_slabs) { lp = ss_off[ls]; lp_end = lp + ss_used[ls]; }
            else { lp = total_ss; lp_end = total_ss + 1; break; }
        }
        const uint4 q = stream[(lp * kWaves + wave) * 64 + lane];
        if (ls < n_slabs) ++lp;
        return q;
    };
    unsigned long long kept = 0;
    int64_t s = blockIdx.x;
    if (s < n_slabs) {
        uint4 e_cur = next_chunk();
        double v[4] = {logp[e_cur.x >> kRowBits], logp[e_cur.y >> kRowBits], logp[e_cur.z >> kRowBits], logp[e_cur.w >> kRowBits]};
        uint4 e_next = next_chunk();
        int rem = ss_used[s];
        for (;;) {
            while (rem == 0) { // slab finished (or without entries): every wave folds ITS 1024 rows, then the next slab
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll 4
                for (int i = 0; i < 16; ++i) {
                    const int rl = wave * 1024 + i * 64 + lane;
                    const double sv = sums[rl];
                    sums[rl] = 0.0;
                    const int64_t r = s * kSlabRows + rl;
                    if (r < n_rows) { row_out[r] = sv; kept += sv >= log_cut ? 1 : 0; }
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                s += G;
                if (s >= n_slabs) break;
                rem = ss_used[s];
            }
            if (s >= n_slabs) break;
            const uint4 e_add = e_cur;
            const double a[4] = {v[0], v[1], v[2], v[3]};
            e_cur = e_next;
            v[0] = logp[e_cur.x >> kRowBits]; v[1] = logp[e_cur.y >> kRowBits]; v[2] = logp[e_cur.z >> kRowBits]; v[3] = logp[e_cur.w >> kRowBits];
            e_next = next_chunk();
            const uint32_t r4[4] = {e_add.x & kRowMask, e_add.y & kRowMask, e_add.z & kRowMask, e_add.w & kRowMask};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ATOMIC) __hip_atomic_fetch_add(&sums[r4[j]], a[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else sums[r4[j]] = sums[r4[j]] + a[j]; // a band holds a row at most once: nobody else touches this slot before the barrier
                if ((j + 1) % GPB == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            --rem;
        }
    }
    if (kept) atomicAdd(kept_out, kept);
}

int main(int argc, char** argv) {
    const int64_t R = argc > 1 ? atoll(argv[1]) : 100000000LL;
    const uint32_t E = argc > 2 ? (uint32_t)atoi(argv[2]) : 100000u;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    std::vector<int64_t> rp((size_t)R + 1);
    rp[0] = 0;
    for (int64_t r = 0; r < R; ++r) rp[r + 1] = rp[r] + 2 + (int64_t)(mix64((uint64_t)r ^ 0xABCDEFULL) % 9);
    const int64_t nnz = rp[R];
    int64_t* d_rp; int32_t* d_col; double *d_logp, *d_ref, *d_out;
    CK(hipMalloc(&d_rp, (size_t)(R + 1) * 8)); CK(hipMalloc(&d_col, (size_t)nnz * 4));
    CK(hipMalloc(&d_logp, (size_t)E * 8)); CK(hipMalloc(&d_ref, (size_t)R * 8)); CK(hipMalloc(&d_out, (size_t)R * 8));
    CK(hipMemcpy(d_rp, rp.data(), (size_t)(R + 1) * 8, hipMemcpyHostToDevice));
    std::vector<double> lp(E);
    for (uint32_t j = 0; j < E; ++j) lp[j] = -18.4 * ((double)(mix64(j * 31u + 7u) >> 11) * 0x1.0p-53) - 1e-3;
    CK(hipMemcpy(d_logp, lp.data(), (size_t)E * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill_cols, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, 0, d_rp, d_col, R, E);
    hipLaunchKernelGGL(k_ref, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, 0, d_rp, d_col, d_logp, R, d_ref);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    // ---- build
    CK(hipEventRecord(e0));
    const int64_t n_slabs = (R + kSlabRows - 1) / kSlabRows;
    int64_t* d_ss; CK(hipMalloc(&d_ss, (size_t)(n_slabs + 1) * 8));
    CK(hipMemset(d_ss, 0, 8));
    hipLaunchKernelGGL(k_alloc, dim3((unsigned)((n_slabs + 255) / 256)), dim3(256), 0, 0, d_rp, R, n_slabs, 10, d_ss + 1);
    std::vector<int64_t> ss((size_t)n_slabs + 1);
    CK(hipMemcpy(ss.data(), d_ss, (size_t)(n_slabs + 1) * 8, hipMemcpyDeviceToHost));
    for (int64_t s = 0; s < n_slabs; ++s) ss[s + 1] += ss[s];
    const int64_t total_ss = ss[n_slabs];
    CK(hipMemcpy(d_ss, ss.data(), (size_t)(n_slabs + 1) * 8, hipMemcpyHostToDevice));
    uint32_t *d_tmp, *d_stream; uint16_t* d_cnt; int* d_ovf;
    CK(hipMalloc(&d_tmp, (size_t)total_ss * 4 * 1024 * 4)); CK(hipMalloc(&d_stream, (size_t)(total_ss + 1) * 4 * 1024 * 4));
    CK(hipMalloc(&d_cnt, (size_t)(total_ss + 1) * 4 * 2)); CK(hipMalloc(&d_ovf, 4)); CK(hipMemset(d_ovf, 0, 4));
    CK(hipMemset(d_cnt, 0, (size_t)(total_ss + 1) * 4 * 2)); // (the no-op superstep: four empty bands)
    hipLaunchKernelGGL(k_bs_place, dim3((unsigned)n_slabs), dim3(256), 0, 0, d_rp, d_col, R, E, d_ss, d_tmp, d_cnt, d_ovf);
    hipLaunchKernelGGL(k_bs_sort, dim3((unsigned)((total_ss + 1) * 4 / GPB)), dim3(256), 0, 0, d_tmp, d_cnt, (total_ss + 1) * 4 / GPB, d_stream);
    CK(hipGetLastError());
    // supersteps actually used per slab
    std::vector<uint16_t> cnt((size_t)total_ss * 4);
    CK(hipMemcpy(cnt.data(), d_cnt, (size_t)total_ss * 4 * 2, hipMemcpyDeviceToHost));
    int ovf = 0; CK(hipMemcpy(&ovf, d_ovf, 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> used((size_t)n_slabs);
    int64_t used_total = 0, entries = 0;
    for (int64_t s = 0; s < n_slabs; ++s) {
        int last = -1;
        for (int64_t b = ss[s] * 4 / GPB; b < ss[s + 1] * 4 / GPB; ++b) { if (cnt[b]) last = (int)(b - ss[s] * 4 / GPB); entries += cnt[b]; }
        used[s] = ((last + 1) * GPB + 3) / 4;
        used_total += used[s];
    }
    int32_t* d_used; CK(hipMalloc(&d_used, (size_t)n_slabs * 4));
    CK(hipMemcpy(d_used, used.data(), (size_t)n_slabs * 4, hipMemcpyHostToDevice));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float build_ms; CK(hipEventElapsedTime(&build_ms, e0, e1));
    printf("# %s: %lld rows, E = %u, nnz %lld; %lld slabs, %lld supersteps allotted, %lld used (%.1f %% slot padding), overflow %d, entries placed %lld, build %.0f ms\n",
           prop.name, (long long)R, E, (long long)nnz, (long long)n_slabs, (long long)total_ss, (long long)used_total,
           100.0 * (1.0 - (double)nnz / ((double)used_total * 4 * 1024)), ovf, (long long)entries, build_ms);
    if (ovf || entries != nnz) { printf("layout failed\n"); return 1; }
    unsigned long long* d_kept; CK(hipMalloc(&d_kept, 8));
    const size_t lds = (size_t)(kSlabRows + 64) * 8;
    const double log_cut = -18.42;
    auto run = [&](const char* what, auto kern) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            CK(hipMemset(d_kept, 0, 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(n_cu < n_slabs ? n_cu : (unsigned)n_slabs), dim3(1024), lds, 0, reinterpret_cast<const uint4*>(d_stream), d_ss, d_used, total_ss, d_logp, R,
                               n_slabs, log_cut, d_out, d_kept);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        CK(hipGetLastError());
        // bit-exact check
        std::vector<double> a((size_t)R), b((size_t)R);
        CK(hipMemcpy(a.data(), d_out, (size_t)R * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), d_ref, (size_t)R * 8, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int64_t r = 0; r < R; ++r) bad += memcmp(&a[r], &b[r], 8) != 0;
        unsigned long long kept = 0; CK(hipMemcpy(&kept, d_kept, 8, hipMemcpyDeviceToHost));
        printf("%-40s %.3f ms per pass (s_i written), rows that differ from the lane-per-row sums: %lld, kept %llu\n", what, best, (long long)bad, kept);
    };
    run("k_bs, ds_add_f64", k_bs<true>);
    run("k_bs, read + add + write", k_bs<false>);
    return 0;
}

// proto_ls.hip -- prototype of a LENGTH-SORTED ROW STREAM for the quantised filter (pass A of FILTER_REFINE).
//
// k_fr_filter_q8o reads a column-sorted slab stream of 3-byte entries (row 13 bits | column offset 11 bits) and scatters one
// LDS atomic per entry into the slab's row sums. The filter's sums are small integers: the order of a row's entries is free,
// and so is the order of the ROWS as long as the list names them. This layout turns the slab around:
//   * the 8192 rows of a slab are sorted by length and cut into 128 items of 64 rows (one row per lane); an item of length
//     L is L "entry rows" (the lane's k-th column, k < L) -- no row ids in the stream at all;
//   * a lane's columns are a bit stream of BITS = 17 (16 for E <= 65536) bits per entry, 32 entry rows = BITS dwords per
//     lane = one UNIT of BITS 256-byte lines: 2.125 bytes per entry instead of 3.03;
//   * the items of wave w (sorted items w, w + 16, ...: every wave gets every length class) of SS consecutive slabs (a
//     "step") are contiguous: a wave streams ~48 SS entry rows per step, one workgroup barrier per step;
//   * per entry: one bit-field extract -> one ds_read_u8 of the 8-bit table in LDS -> a register add. No LDS atomics, no
//     row sums in LDS; at an item's end (a bit mask per unit says where) one ballot; the lists hold POSITIONS
//     (j * 1024 + wave * 64 + lane), the exact pass translates them through a u16 permutation.
// This file: synthetic stream of config 3's shape built directly in the layout, the kernel, a check kernel, timing, and a
// probe of what the LDS gives random byte reads.
// (v1 -- 16-bit halves + a lane mask per entry row for bit 16, units of 16 rows, a barrier per slab, per-row scalar
// bookkeeping: 0.366 ms, 0.307 without the masks: instruction-bound, not HBM-bound.)
//   hipcc --offload-arch=gfx950 -O3 -o benchmarks/bin/proto_ls benchmarks/proto_ls.hip && ./benchmarks/bin/proto_ls [slabs]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define AS1 __attribute__((address_space(1)))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __host__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31);
}
#ifndef BITS
#define BITS 17
#endif
#ifndef SS
#define SS 4 // list slabs per step
#endif
#ifndef ABL
#define ABL 0 // 1: table reads at conflict-free addresses; 2: no list emission
#endif
constexpr int kSlabRows = 8192, kWaves = 16, kIPW = 8, kUnitRows = 32, kThreads = kWaves * 64, kItems = kIPW * SS;
constexpr uint32_t kColMask = (1u << BITS) - 1u;

struct RangeDesc { // one per (step, wave): 16 bytes
    uint32_t line0;  // first 256-byte line of the range
    uint32_t n_rows; // entry rows of the range
    uint32_t bm0;    // index of its first unit in the boundary-mask array
    uint32_t pad;
};

__host__ __device__ inline int item_len(int wave, int j) { return 2 + ((wave + 16 * j) * 9) / 128; }

// one thread per (range, lane): the lane's bit stream
__global__ void k_fill(const RangeDesc* __restrict__ desc, int64_t n_ranges, uint32_t E, uint32_t* __restrict__ ent) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(id & 63);
    const int64_t rg = id >> 6;
    if (rg >= n_ranges) return;
    const RangeDesc d = desc[rg];
    uint64_t acc = 0;
    int nb = 0;
    uint32_t dw = 0;
    for (uint32_t t = 0; t < d.n_rows; ++t) {
        const uint32_t col = (uint32_t)(mix64((((uint64_t)rg << 20) | ((uint64_t)t << 6) | (uint64_t)lane) * 0x9E3779B1ULL + 12345) % E);
        acc |= (uint64_t)col << nb;
        nb += BITS;
        if (nb >= 32) { ent[((size_t)d.line0 + dw) * 64 + lane] = (uint32_t)acc; acc >>= 32; nb -= 32; ++dw; }
    }
    if (nb > 0) ent[((size_t)d.line0 + dw) * 64 + lane] = (uint32_t)acc;
}

__device__ inline uint32_t get_col(const uint32_t* __restrict__ ent, uint32_t line0, uint32_t t, int lane) {
    const uint32_t o = t * BITS, w = o >> 5, sh = o & 31;
    uint64_t x = ent[((size_t)line0 + w) * 64 + lane];
    if (sh + BITS > 32) x |= (uint64_t)ent[((size_t)line0 + w + 1) * 64 + lane] << 32;
    return (uint32_t)(x >> sh) & kColMask;
}

// the sums recomputed plainly: one thread per (range, item, lane)
__global__ void k_check(const RangeDesc* __restrict__ desc, int64_t n_ranges, const uint32_t* __restrict__ ent, const uint8_t* __restrict__ tab, int thr,
                        unsigned long long* __restrict__ n_cand, unsigned* __restrict__ per_slab, unsigned long long* __restrict__ dbg) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = (int)(id & 63);
    const int jj = (int)((id >> 6) % kItems);
    const int64_t rg = (id >> 6) / kItems;
    if (rg >= n_ranges) return;
    const RangeDesc d = desc[rg];
    const int wave = (int)(rg % kWaves);
    uint32_t t = 0;
    for (int q = 0; q < jj; ++q) t += (uint32_t)item_len(wave, q % kIPW);
    const int L = item_len(wave, jj % kIPW);
    if (t + L > d.n_rows) return; // (the last step's missing slabs)
    unsigned sum = 0;
    for (int k = 0; k < L; ++k) sum += tab[get_col(ent, d.line0, t + k, lane)];
    if ((int)sum <= thr) {
        atomicAdd(n_cand, 1ULL);
        atomicAdd(&per_slab[(rg / kWaves) * SS + jj / kIPW], 1u);
        if (dbg) atomicOr(&dbg[rg * kItems + jj], 1ULL << lane);
    }
}

// ---- the kernel ----------------------------------------------------------------------------------------------------------
struct Unit {
    uint32_t d[BITS];
    uint32_t bm;
};

__device__ __forceinline__ void ls_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename T>
__device__ __forceinline__ const T AS1* uniform_ptr(const T* p) {
    const uint64_t v = (uint64_t)p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return (const T AS1*)(((uint64_t)hi << 32) | lo);
}

// a unit: BITS lines of 256 bytes + its boundary mask (a vector load of one address: scalar loads would share lgkmcnt with
// the LDS reads)
__device__ __forceinline__ void load_unit(Unit& U, const uint32_t* __restrict__ ent, const uint32_t* __restrict__ bm, uint32_t line, uint32_t bmi, int lane,
                                          uint32_t zero) {
    const uint32_t AS1* p = uniform_ptr(ent + (size_t)line * 64) + (uint32_t)lane;
#pragma unroll
    for (int q = 0; q < BITS; ++q) U.d[q] = p[q * 64];
    const char AS1* bp = reinterpret_cast<const char AS1*>(uniform_ptr(bm + bmi));
    U.bm = *reinterpret_cast<const uint32_t AS1*>(bp + zero);
}

__device__ __forceinline__ RangeDesc load_desc(const RangeDesc* __restrict__ desc, int64_t idx, uint32_t zero) {
    const char AS1* bp = reinterpret_cast<const char AS1*>(uniform_ptr(desc + idx));
    const u32x4 v = *reinterpret_cast<const u32x4 AS1*>(bp + zero);
    RangeDesc d;
    d.line0 = v.x; d.n_rows = v.y; d.bm0 = v.z; d.pad = v.w;
    return d;
}
__device__ __forceinline__ void desc_uniform(RangeDesc& d) {
    d.line0 = __builtin_amdgcn_readfirstlane(d.line0); d.n_rows = __builtin_amdgcn_readfirstlane(d.n_rows);
    d.bm0 = __builtin_amdgcn_readfirstlane(d.bm0);
}

__global__ __launch_bounds__(kThreads) void k_ls_filter(const uint32_t* __restrict__ ent, const uint32_t* __restrict__ bm,
                                                         const RangeDesc* __restrict__ desc, const uint8_t* __restrict__ q8, int tab_bytes,
                                                         int64_t n_steps, int64_t n_slabs, uint32_t dummy_line, uint32_t dummy_bm, int threshold,
                                                         uint16_t* __restrict__ cand_list, unsigned* __restrict__ cand_cnt,
                                                         unsigned long long* __restrict__ cand_count, unsigned long long* __restrict__ dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* s_m = reinterpret_cast<unsigned long long*>(smem); // [2][SS][128] ballots, index (ls * 8 + j) * 16 + wave
    uint8_t* tab = reinterpret_cast<uint8_t*>(smem + 2 * SS * 128 * 8);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t zero_v = 0u;
    asm volatile("" : "+v"(zero_v));
    const int64_t G = gridDim.x;
    const int64_t q_wg = n_steps / G, r_wg = n_steps % G, b_wg = blockIdx.x;
    const int64_t s_begin = b_wg * q_wg + (b_wg < r_wg ? b_wg : r_wg), s_end = s_begin + q_wg + (b_wg < r_wg ? 1 : 0);
    const int64_t last = n_steps * kWaves; // desc[last]: an empty range
    auto didx = [&](int64_t s) { return s < s_end ? s * kWaves + wave : last; };
    // descriptors run two steps ahead of the adds (their loads must be OLDER than a whole step's unit loads)
    RangeDesc d_cur = load_desc(desc, didx(s_begin), zero_v);
    RangeDesc d_nxt = load_desc(desc, didx(s_begin + 1), zero_v);
    desc_uniform(d_cur);
    Unit A, B;
    load_unit(A, ent, bm, d_cur.n_rows > 0 ? d_cur.line0 : dummy_line, d_cur.n_rows > 0 ? d_cur.bm0 : dummy_bm, lane, zero_v);
    {
        const uint4* src = reinterpret_cast<const uint4*>(q8);
        uint4* dst = reinterpret_cast<uint4*>(tab);
        for (int i = tid; i < tab_bytes / 16; i += kThreads) dst[i] = src[i];
    }
    for (int i = tid; i < 2 * SS * 128; i += kThreads) s_m[i] = 0ULL;
    ls_barrier();
    unsigned long long n_cand = 0;
    int it = 0;
    int64_t s_emit = -1;
    for (int64_t s = s_begin;; ++s, ++it) {
        const bool have = s < s_end;
        RangeDesc d_nn = load_desc(desc, didx(s + 2), zero_v);
        if (ABL != 2 && s_emit >= 0) { // ---- the lists of the previous step's slabs: (j, wave, lane) order
            unsigned long long* mall = s_m + ((it - 1) & 1) * (SS * 128);
#pragma unroll 1
            for (int ls = 0; ls < SS; ++ls) {
                const int64_t slab = s_emit * SS + ls;
                if (slab >= n_slabs) break;
                unsigned long long* mm = mall + ls * 128;
                const unsigned c0 = (unsigned)__builtin_popcountll(mm[2 * lane]), c1 = (unsigned)__builtin_popcountll(mm[2 * lane + 1]);
                unsigned incl = c0 + c1;
#pragma unroll
                for (int dd = 1; dd < 64; dd <<= 1) {
                    const unsigned o = __shfl_up(incl, dd, 64);
                    if (lane >= dd) incl += o;
                }
                const unsigned ex0 = incl - c0 - c1, ex1 = ex0 + c0;
                const unsigned total = __shfl(incl, 63, 64);
                uint16_t* seg = cand_list + slab * (int64_t)kSlabRows;
#pragma unroll
                for (int j = 0; j < kIPW; ++j) {
                    const int idx = j * kWaves + wave;
                    const unsigned long long mv = mm[idx]; // (one address: a broadcast read)
                    const unsigned long long m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(mv >> 32)) << 32) |
                                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)mv);
                    if (m != 0ULL) {
                        const unsigned off = __shfl((idx & 1) ? ex1 : ex0, idx >> 1, 64);
                        if ((m >> lane) & 1ULL) {
                            const unsigned pos = off + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                            seg[pos] = (uint16_t)(j * 1024 + tid);
                        }
                    }
                }
                if (tid == 0) cand_cnt[slab] = total;
                n_cand += total;
            }
        }
        if (have) {
            unsigned long long* mw = s_m + (it & 1) * (SS * 128) + wave;
            const int n_units = (int)((d_cur.n_rows + kUnitRows - 1) / kUnitRows);
            const uint32_t line0 = d_cur.line0, bm0 = d_cur.bm0;
            int jj = 0;
            unsigned sum = 0;
#define LS_HALF(UC, UN, UI)                                                                                                  \
            {                                                                                                                \
                const bool more = (UI) + 1 < n_units;                                                                        \
                load_unit(UN, ent, bm, more ? line0 + (uint32_t)((UI) + 1) * BITS : dummy_line, more ? bm0 + (uint32_t)(UI) + 1u : dummy_bm, lane, zero_v); \
                unsigned v[kUnitRows];                                                                                       \
                _Pragma("unroll") for (int t = 0; t < kUnitRows; ++t) {                                                      \
                    const int o = t * BITS, w = o >> 5, sh = o & 31;                                                         \
                    uint32_t col;                                                                                            \
                    if (sh + BITS <= 32) col = __builtin_amdgcn_ubfe(UC.d[w], (uint32_t)sh, (uint32_t)BITS);                 \
                    else col = __builtin_amdgcn_alignbit(UC.d[w + 1 < BITS ? w + 1 : w], UC.d[w], (uint32_t)sh) & kColMask;  \
                    v[t] = ABL == 1 ? tab[(col & 0x1ff00u) + (uint32_t)lane] : tab[col];                                     \
                }                                                                                                            \
                uint32_t bmv = (uint32_t)__builtin_amdgcn_readfirstlane((int)UC.bm);                                         \
                _Pragma("unroll") for (int t = 0; t < kUnitRows; ++t) {                                                      \
                    sum += v[t];                                                                                             \
                    if (__builtin_expect((bmv >> t) & 1u, 0)) {                                                              \
                        const unsigned long long m = __ballot((int)sum <= threshold);                                        \
                        if (lane == 0) mw[jj * kWaves] = m;                                                                  \
                        if (dbg && lane == 0) dbg[(s * kWaves + wave) * kItems + jj] = m;                                    \
                        sum = 0;                                                                                             \
                        jj = __builtin_amdgcn_readfirstlane(jj + 1);                                                         \
                    }                                                                                                        \
                }                                                                                                            \
            }
            for (int u = 0; u < n_units;) {
                LS_HALF(A, B, u)
                ++u;
                if (u >= n_units) break;
                LS_HALF(B, A, u)
                ++u;
            }
#undef LS_HALF
            // (the masks of slabs the last step does not have stay as they were: never read)
        }
        // the next step's first unit goes out before this step's barrier
        desc_uniform(d_nxt);
        load_unit(A, ent, bm, d_nxt.n_rows > 0 ? d_nxt.line0 : dummy_line, d_nxt.n_rows > 0 ? d_nxt.bm0 : dummy_bm, lane, zero_v);
        d_cur = d_nxt;
        d_nxt = d_nn;
        s_emit = have ? s : -1;
        if (!have) break;
        ls_barrier();
    }
    if (tid == 0 && n_cand) atomicAdd(cand_count, n_cand);
}

// ---- what the LDS gives random byte reads: 16 waves, 8 independent reads per lane and trip ---------------------------------
template <bool RANDOM>
__global__ __launch_bounds__(kThreads) void k_lds_probe(const uint8_t* __restrict__ q8, int tab_bytes, int trips, unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint8_t* tab = reinterpret_cast<uint8_t*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < tab_bytes / 16; i += kThreads) reinterpret_cast<uint4*>(tab)[i] = reinterpret_cast<const uint4*>(q8)[i];
    __syncthreads();
    uint32_t x = (uint32_t)mix64((uint64_t)tid * 977 + blockIdx.x);
    unsigned sum = 0;
    for (int i = 0; i < trips; ++i) {
        unsigned v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x = x * 1664525u + 1013904223u;
            const uint32_t a = RANDOM ? (x >> 8) % 100000u : (((x >> 8) % 100000u) & 0x1ff00u) + (uint32_t)lane;
            v[k] = tab[a];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
    }
    if (sum == 0x12345678u) out[0] = sum;
}

int main(int argc, char** argv) {
    const int64_t n_slabs = argc > 1 ? atoll(argv[1]) : 12207;
    const int64_t n_steps = (n_slabs + SS - 1) / SS;
    const uint32_t E = BITS == 16 ? 65000 : 100000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const int64_t n_ranges = n_steps * kWaves;
    std::vector<RangeDesc> h_desc((size_t)n_ranges + 1);
    std::vector<uint32_t> h_bm;
    uint64_t line = 0, n_entries = 0;
    for (int64_t s = 0; s < n_steps; ++s)
        for (int w = 0; w < kWaves; ++w) {
            RangeDesc d{};
            d.line0 = (uint32_t)line;
            d.bm0 = (uint32_t)h_bm.size();
            uint32_t n = 0;
            std::vector<uint32_t> ends;
            for (int ls = 0; ls < SS && s * SS + ls < n_slabs; ++ls)
                for (int j = 0; j < kIPW; ++j) { n += (uint32_t)item_len(w, j); ends.push_back(n - 1); }
            d.n_rows = n;
            const uint32_t n_units = (n + kUnitRows - 1) / kUnitRows;
            for (uint32_t u = 0; u < n_units; ++u) h_bm.push_back(0u);
            for (uint32_t e : ends) h_bm[d.bm0 + e / kUnitRows] |= 1u << (e % kUnitRows);
            h_desc[(size_t)(s * kWaves + w)] = d;
            n_entries += (uint64_t)n * 64;
            line += ((uint64_t)n * BITS + 31) / 32;
        }
    const uint32_t dummy_line = (uint32_t)line, dummy_bm = (uint32_t)h_bm.size();
    h_bm.push_back(0u);
    h_desc[(size_t)n_ranges] = RangeDesc{dummy_line, 0, dummy_bm, 0};
    const uint64_t total_lines = line + 2 * BITS;
    uint32_t* d_ent; CK(hipMalloc(&d_ent, total_lines * 256)); CK(hipMemset(d_ent, 0, total_lines * 256));
    uint32_t* d_bm; CK(hipMalloc(&d_bm, h_bm.size() * 4)); CK(hipMemcpy(d_bm, h_bm.data(), h_bm.size() * 4, hipMemcpyHostToDevice));
    RangeDesc* d_desc; CK(hipMalloc(&d_desc, h_desc.size() * sizeof(RangeDesc)));
    CK(hipMemcpy(d_desc, h_desc.data(), h_desc.size() * sizeof(RangeDesc), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((n_ranges * 64 + 255) / 256)), dim3(256), 0, 0, d_desc, n_ranges, E, d_ent);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    // the table: levels of log-uniform probabilities on [1e-8, 1e-3] at 224 steps over -log(1e-8); all of 2^BITS addressable
    const int tab_bytes = BITS == 16 ? 65536 : ((int)E + 2 + 15) / 16 * 16;
    std::vector<uint8_t> h_tab((size_t)(1 << BITS), 0);
    for (uint32_t e = 0; e < E; ++e) {
        const double u = (double)(mix64(e * 7919ULL + 3) >> 11) * 0x1.0p-53;
        const double neg = (8.0 - 5.0 * u) * 2.302585092994046;
        const double lv = neg / (18.420680743952367 / 224.0);
        h_tab[e] = (uint8_t)(lv >= 255.0 ? 255 : (int)lv);
    }
    h_tab[E] = 0; h_tab[E + 1] = 255;
    uint8_t* d_tab; CK(hipMalloc(&d_tab, h_tab.size())); CK(hipMemcpy(d_tab, h_tab.data(), h_tab.size(), hipMemcpyHostToDevice));
    uint16_t* d_list; CK(hipMalloc(&d_list, (size_t)n_steps * SS * kSlabRows * 2));
    unsigned* d_cnt; CK(hipMalloc(&d_cnt, (size_t)n_steps * SS * 4)); CK(hipMemset(d_cnt, 0, (size_t)n_steps * SS * 4));
    unsigned* d_cnt_ref; CK(hipMalloc(&d_cnt_ref, (size_t)n_steps * SS * 4)); CK(hipMemset(d_cnt_ref, 0, (size_t)n_steps * SS * 4));
    unsigned long long* d_tot; CK(hipMalloc(&d_tot, 16)); CK(hipMemset(d_tot, 0, 16));
    const int thr = 224;
    unsigned long long *d_dbg = nullptr, *d_dbg2 = nullptr;
    const bool debug = n_slabs <= 2000;
    if (debug) { CK(hipMalloc(&d_dbg, (size_t)n_ranges * kItems * 8)); CK(hipMalloc(&d_dbg2, (size_t)n_ranges * kItems * 8)); CK(hipMemset(d_dbg, 0, (size_t)n_ranges * kItems * 8)); CK(hipMemset(d_dbg2, 0, (size_t)n_ranges * kItems * 8)); }
    const size_t lds = 2 * SS * 128 * 8 + (size_t)tab_bytes;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ls_filter), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (int)(n_steps < n_cu ? n_steps : n_cu);
    const double bytes = (double)total_lines * 256.0 + (double)n_ranges * 16 + (double)h_bm.size() * 4;
    printf("# %s, %d CUs; BITS %d, SS %d, ABL %d; %lld slabs, %.4g entries; stream %.3f GB (%.3f B per entry); table %d B; grid %d\n", prop.name, n_cu, BITS, SS, ABL,
           (long long)n_slabs, (double)n_entries, bytes * 1e-9, bytes / (double)n_entries, tab_bytes, grid);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipMemsetAsync(d_tot, 0, 8));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_ls_filter, dim3(grid), dim3(kThreads), lds, 0, d_ent, d_bm, d_desc, d_tab, tab_bytes, n_steps, n_slabs, dummy_line, dummy_bm, thr,
                           d_list, d_cnt, d_tot, d_dbg);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 1 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    unsigned long long tot = 0, tot_ref = 0;
    CK(hipMemcpy(&tot, d_tot, 8, hipMemcpyDeviceToHost));
    hipLaunchKernelGGL(k_check, dim3((unsigned)((n_ranges * kItems * 64 + 255) / 256)), dim3(256), 0, 0, d_desc, n_ranges, d_ent, d_tab, thr, d_tot + 1, d_cnt_ref, d_dbg2);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    CK(hipMemcpy(&tot_ref, d_tot + 1, 8, hipMemcpyDeviceToHost));
    std::vector<unsigned> c((size_t)n_slabs), cr((size_t)n_slabs);
    CK(hipMemcpy(c.data(), d_cnt, (size_t)n_slabs * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(cr.data(), d_cnt_ref, (size_t)n_slabs * 4, hipMemcpyDeviceToHost));
    int64_t bad = 0;
    for (int64_t s = 0; s < n_slabs; ++s) {
        if (c[(size_t)s] != cr[(size_t)s] && bad < 6) printf("  slab %lld: %u, check %u\n", (long long)s, c[(size_t)s], cr[(size_t)s]);
        bad += c[(size_t)s] != cr[(size_t)s];
    }
    std::vector<uint16_t> l0(kSlabRows);
    CK(hipMemcpy(l0.data(), d_list + (n_slabs - 1) * kSlabRows, kSlabRows * 2, hipMemcpyDeviceToHost));
    bool asc = true;
    for (unsigned i = 1; i < c[(size_t)n_slabs - 1]; ++i) asc &= l0[i] > l0[i - 1];
    printf("filter %.4f ms  -> %.3f TB/s; candidates %llu (check %llu), slabs with a different count %lld, last list ascending %d (%u)\n", best,
           bytes / (best * 1e-3) * 1e-12, tot, tot_ref, (long long)bad, (int)asc, c[(size_t)n_slabs - 1]);
    if (debug) {
        std::vector<unsigned long long> a((size_t)n_ranges * kItems), b((size_t)n_ranges * kItems);
        CK(hipMemcpy(a.data(), d_dbg, a.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), d_dbg2, b.size() * 8, hipMemcpyDeviceToHost));
        int shown = 0;
        for (size_t i = 0; i < a.size() && shown < 12; ++i)
            if (a[i] != b[i]) { printf("  range %zu (step %zu wave %zu) item %zu: kernel %016llx check %016llx\n", i / kItems, i / kItems / kWaves, (i / kItems) % kWaves, i % kItems, a[i], b[i]); ++shown; }
    }
    const bool ok = ABL != 0 || (tot == tot_ref && bad == 0 && asc);
    // the LDS probe
    if (argc > 2) {
        const int trips = 4096;
        for (int rnd = 0; rnd < 2; ++rnd) {
            auto kern = rnd ? k_lds_probe<true> : k_lds_probe<false>;
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 100016));
            float bt = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(kern, dim3(n_cu), dim3(kThreads), 100016, 0, d_tab, 100016, trips, d_cnt);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < bt) bt = ms;
            }
            CK(hipGetLastError());
            const double instr_per_cu = (double)trips * 8 * kWaves;
            printf("LDS probe, %s byte reads: %.3f ms for %.0f ds_read_u8 per CU -> %.2f ns each (%.1f clk at 2.1 GHz); config 3's 9.42e6 entry rows would take %.3f ms\n",
                   rnd ? "random" : "conflict-free", bt, instr_per_cu, bt * 1e6 / instr_per_cu, bt * 1e6 / instr_per_cu * 2.1, 9.42e6 / n_cu * bt / instr_per_cu);
        }
    }
    return ok ? 0 : 1;
}

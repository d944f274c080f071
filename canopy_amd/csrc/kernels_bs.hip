// kernels_bs.hip -- the "band stream" (BS) variant of K1..K4: exact CSR-order fp64 row sums with WORKGROUP-wide slabs,
// gfx950 / wave64 only (round 3).
//
// Why. K1 as literally stated is s_i = the sequential fp64 sum of a row's logp values in CSR (= ascending column) order. The
// wave-slab stream (kernels_ws.hip) gets that order for free -- a wave owns 4096 rows and its LDS adds execute in issue
// order -- but one sorted stream per WAVE fixes the density of its table gathers: 24.5 K entries over E = 1e5 columns are
// ~17 cache lines and ~45 texture-addresser clocks per 64-lane gather, one wave per SIMD, 1.1-1.2 ms per pass at config 3.
// A slab of 16384 rows shared by the 16 waves of a workgroup has a four times denser stream (~6 lines, ~24 clocks) and
// sixteen waves to hide latencies with; what it needs is another way to keep the order:
//
//   * the slab's entries are cut into BANDS of 2048 slots (16 waves x 2 groups of 64) such that EVERY ROW APPEARS AT MOST
//     ONCE IN A BAND and a row's entries sit in strictly increasing bands in CSR order. An entry's band is its column's share
//     of the slab's nominal band count, pushed later while the row's previous entry sits in the same or a later band or the
//     band is full (the greedy of the wave-slab builder, with bands for groups);
//   * inside a band the entries are sorted by column and dealt to the waves in runs of 128, so a wave's gather instruction
//     covers 64 consecutive columns' worth of entries of the WHOLE slab;
//   * the kernel adds a band's entries into the slab's 16384 fp64 LDS sums -- all rows of a band are distinct, so no two
//     lanes of the workgroup touch the same sum -- and a workgroup barrier (LDS drained first) separates the bands: row i
//     receives logp[c0], logp[c1], ... in CSR order, the same chain of IEEE additions as in every other exact kernel and in
//     the oracle -> bit-identical s_i;
//   * stream loads run two supersteps (a superstep = one 16-byte load per lane = 4 groups = 2 bands) ahead of the adds,
//     the gathers one superstep ahead, across slab boundaries; a slab ends with every wave folding ITS 1024 rows
//     (ws_epilogue) between two barriers.
//
// Measured at config 3: 0.93 ms per pass with every s_i written against 1.19 for the wave-slab stream (DESIGN.md 3;
// benchmarks/microbench8.hip: the bare loop; benchmarks/proto_bs.hip: the standalone prototype -- bands of 1024 / 4096 slots
// 1.01 / 0.98 there; benchmarks/build_bs_variant.sh builds the library with another geometry: BS_ROWS, BS_WAVES, BS_GPB,
// BS_WGS -- two workgroups of 8 waves on 8192-row slabs per CU: 1.07).
// The placement uses LDS atomics: WHICH slot of a band an entry takes can differ from build to build, the sums cannot.
#include <stdlib.h>

#include <vector>

#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

#ifndef BS_ROWS
#define BS_ROWS 16384
#endif
#ifndef BS_WAVES
#define BS_WAVES 16
#endif
#ifndef BS_GPB
#define BS_GPB 2
#endif
#ifndef BS_WGS
#define BS_WGS 1 // workgroups per CU
#endif
constexpr int kBsSlabRows = BS_ROWS, kBsWaves = BS_WAVES, kBsThreads = kBsWaves * kWave;
constexpr int kBsGroupsPerBand = BS_GPB;                  // per wave: a barrier follows every BS_GPB adds
constexpr int kBsBand = kWave * kBsWaves * kBsGroupsPerBand; // slots per band
constexpr int kBsFill = kBsBand * 995 / 1024;             // nominal entries per band (slack for the column shares' scatter)
constexpr int kBsRowBits = BS_ROWS > 8192 ? 15 : 14;      // the slab's rows + 64 sink slots
constexpr uint32_t kBsRowMask = (1u << kBsRowBits) - 1;
constexpr int kBsMaxBands = 8192;                         // per slab (LDS counters of the placement)
constexpr int kBsSsEntries = kBsThreads * 4;              // entries per superstep (16 KiB)
constexpr int kBsBandsPerSs = 4 / kBsGroupsPerBand;
constexpr int kBsPend = 128;                              // per wave: kept s_i waiting for a full batch of 64 (exp / log1p)
constexpr size_t kBsSumsBytes = (size_t)(kBsSlabRows + kWave) * 8;
constexpr size_t kBsLdsBytes = kBsSumsBytes + (size_t)kBsWaves * kBsPend * 8;

__device__ __forceinline__ int64_t bs_uload(const int64_t* __restrict__ p) {
    const int64_t v = *p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | (uint64_t)lo);
}

__device__ __forceinline__ void bs_barrier() { // LDS drained, then the workgroup barrier
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- builder ------------------------------------------------------------------------------------------------------
// supersteps allotted to a slab: nominal bands + an eighth + one per entry of the slab's longest row (a row of L entries
// needs L different bands) + 4. One workgroup per slab (the longest row of THIS slab: one row of thousands of entries
// somewhere in a large model must not inflate every slab's allotment).
__global__ __launch_bounds__(256) void k_bs_alloc(const int64_t* __restrict__ row_ptr, int64_t n_rows, int64_t* __restrict__ ss,
                                                  int* __restrict__ too_many) {
    __shared__ int s_max;
    const int64_t s = blockIdx.x;
    const int64_t r0 = s * kBsSlabRows, r1 = r0 + kBsSlabRows < n_rows ? r0 + kBsSlabRows : n_rows;
    if (threadIdx.x == 0) s_max = 0;
    __syncthreads();
    int64_t mx = 0;
    for (int64_t r = r0 + threadIdx.x; r < r1; r += 256) {
        const int64_t len = row_ptr[r + 1] - row_ptr[r];
        mx = len > mx ? len : mx;
    }
    atomicMax(&s_max, (int)(mx > (int64_t)kBsMaxBands ? (int64_t)kBsMaxBands : mx));
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int64_t nnz = row_ptr[r1] - row_ptr[r0];
    int64_t nom = (nnz + kBsFill - 1) / kBsFill;
    if (nom < 1) nom = 1;
    const int64_t alloc = nom + nom / 8 + s_max + 4;
    if (alloc > kBsMaxBands) atomicExch(too_many, 1);
    ss[s] = (alloc + kBsBandsPerSs - 1) / kBsBandsPerSs;
}

// phase 1: one workgroup per slab, band counters in LDS; the entries land in tmp[band][arrival order]
__global__ __launch_bounds__(256) void k_bs_place(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col, int64_t n_rows,
                                                  uint32_t n_events, const int64_t* __restrict__ ss_off, uint32_t* __restrict__ tmp,
                                                  uint16_t* __restrict__ band_cnt, int32_t* __restrict__ ss_used, int* __restrict__ overflow) {
    __shared__ unsigned cnt[kBsMaxBands];
    __shared__ int s_last;
    const int64_t s = blockIdx.x;
    const int64_t r0 = s * kBsSlabRows, r1 = r0 + kBsSlabRows < n_rows ? r0 + kBsSlabRows : n_rows;
    int n_alloc = (int)((ss_off[s + 1] - ss_off[s]) * kBsBandsPerSs);
    if (n_alloc > kBsMaxBands) n_alloc = kBsMaxBands; // (k_bs_alloc has flagged the model)
    const int64_t nnz = row_ptr[r1] - row_ptr[r0];
    uint64_t nom = (uint64_t)((nnz + kBsFill - 1) / kBsFill);
    if (nom < 1) nom = 1;
    for (int i = threadIdx.x; i < n_alloc; i += 256) cnt[i] = 0u;
    if (threadIdx.x == 0) s_last = -1;
    __syncthreads();
    uint32_t* out = tmp + ss_off[s] * kBsSsEntries;
    bool ovf = false;
    const int n_nom = (int)nom;
    for (int64_t r = r0 + threadIdx.x; r < r1 && !ovf; r += 256) {
        const int64_t kb = row_ptr[r];
        const int L = (int)(row_ptr[r + 1] - kb);
        // A row whose last entries share the last nominal bands would open new, nearly empty bands behind them (two per
        // slab at config 3: 4 % of the stream, ~1 % of the pass -- sparse bands are cheap): such a row's TRAILING entries are planned backwards from the last
        // nominal band instead (g: the latest strictly increasing bands at or before the nominal ones), its leading entries
        // forwards as usual (f), switching where g passes f. Rows of more than 32 entries keep the plain forward rule.
        int tgt[32];
        const bool planned = L >= 1 && L <= 32;
        if (planned) {
            int f[32], x = -1;
            for (int k = 0; k < L; ++k) {
                tgt[k] = (int)(((uint64_t)(uint32_t)col[kb + k] * nom) / n_events);
                x = tgt[k] > x + 1 ? tgt[k] : x + 1;
                f[k] = x;
            }
            if (f[L - 1] > n_nom - 1 && L <= n_nom) {
                int g[32];
                x = n_nom;
                for (int k = L - 1; k >= 0; --k) { x = tgt[k] < x - 1 ? tgt[k] : x - 1; g[k] = x; }
                int k0 = -1;
                for (int k = 0; k < L && k0 < 0; ++k)
                    if (k == 0 ? g[0] >= 0 : g[k] > f[k - 1]) k0 = k;
                if (k0 >= 0) {
                    // ... and an entry that had to move takes a band up to 7 before its latest one (left at the latest, the
                    // moved entries of all such rows pile into the last two or three bands and overflow them)
                    const int base = k0 > 0 ? f[k0 - 1] + 1 : 0; // entry k0's earliest band; entry k needs k - k0 bands below it
                    x = n_nom;
                    for (int k = L - 1; k >= k0; --k) {
                        const int hi = tgt[k] < x - 1 ? tgt[k] : x - 1;
                        int b = hi;
                        if (hi < tgt[k]) {
                            const int lo = base + (k - k0);
                            b = hi - (int)((uint32_t)((uint64_t)(r * 31 + k) * 0x9E3779B97F4A7C15ULL >> 61)); // 0 .. 7
                            if (b < lo) b = lo;
                        }
                        g[k] = b;
                        x = b;
                    }
                    for (int k = 0; k < L; ++k) tgt[k] = k < k0 ? f[k] : g[k];
                }
            }
        }
        int b_prev = -1;
        for (int64_t k = kb; k < kb + L && !ovf; ++k) {
            const uint32_t c = (uint32_t)col[k];
            int b = planned ? tgt[k - kb] : (int)(((uint64_t)c * nom) / n_events);
            if (b <= b_prev) b = b_prev + 1;
            for (;;) {
                if (b >= n_alloc) { ovf = true; break; }
                const unsigned pos = atomicAdd(&cnt[b], 1u);
                if (pos < (unsigned)kBsBand) { out[(int64_t)b * kBsBand + pos] = (c << kBsRowBits) | (uint32_t)(r - r0); break; }
                ++b;
            }
            b_prev = b;
        }
    }
    if (ovf) atomicExch(overflow, 1);
    __syncthreads();
    int last = -1;
    for (int i = threadIdx.x; i < n_alloc; i += 256) {
        const unsigned c = cnt[i] < (unsigned)kBsBand ? cnt[i] : (unsigned)kBsBand;
        band_cnt[ss_off[s] * kBsBandsPerSs + i] = (uint16_t)c;
        if (c) last = i;
    }
    if (last >= 0) atomicMax(&s_last, last);
    __syncthreads();
    if (threadIdx.x == 0) ss_used[s] = (s_last + 1 + kBsBandsPerSs - 1) / kBsBandsPerSs;
}

// phase 2: one workgroup per band: bitonic sort by column in LDS, then the transposed write -- sorted position q is group
// q / 64; a wave takes kBsGroupsPerBand consecutive groups; stream[superstep][wave][lane][slot & 3], slot = band * groups + sub
__global__ __launch_bounds__(256) void k_bs_sort(const uint32_t* __restrict__ tmp, const uint16_t* __restrict__ band_cnt, int64_t n_bands,
                                                 uint32_t* __restrict__ stream) {
    __shared__ uint32_t key[kBsBand];
    const int64_t b = blockIdx.x;
    if (b >= n_bands) return;
    const int n = band_cnt[b];
    for (int i = threadIdx.x; i < kBsBand; i += 256) key[i] = i < n ? tmp[b * kBsBand + i] : 0xFFFFFFFFu;
    __syncthreads();
    if (n > 1) { // (workgroup-uniform)
        for (int k = 2; k <= kBsBand; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = threadIdx.x; t < kBsBand / 2; t += 256) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)); // the lower index of pair t
                    const int p = i | j;
                    const bool up = (i & k) == 0;
                    const uint32_t a = key[i], c = key[p];
                    if ((a > c) == up) { key[i] = c; key[p] = a; }
                }
                __syncthreads();
            }
    }
    const uint32_t last_col = n > 0 ? (key[n - 1] >> kBsRowBits) : 0u;
    for (int q = threadIdx.x; q < kBsBand; q += 256) {
        const int g = q >> 6, lane = q & 63;
        // padding: the band's last column (a line the band touches anyway) -> one of the 64 sink rows, lane k on slot k
        const uint32_t e = q < n ? key[q] : ((last_col << kBsRowBits) | (uint32_t)(kBsSlabRows + lane));
        const int w = g / kBsGroupsPerBand, sub = g % kBsGroupsPerBand;
        const int64_t t = b * kBsGroupsPerBand + sub;
        stream[(((t >> 2) * kBsWaves + w) * kWave + lane) * 4 + (t & 3)] = e;
    }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------
struct BsScalars {
    int64_t total_ss; // the stream has one no-op superstep after it
    int64_t n_rows, n_slabs, row_offset;
    double log_cut;
    int32_t limit_order;
    int32_t need_len;
};

// The end of a slab: every wave takes ITS 1024 rows out of the sums (filter decision, count, argmax, optional outputs) and
// clears them. Inline, outputs as template parameters: the stores are then ordinary younger vector-memory operations behind
// the gathers already in flight and the loop's next wait stays a vmcnt(N) (behind a call the compiler waits for everything,
// i.e. for the stores to be acknowledged -- with all 16 waves of the CU in the epilogue at once nothing hides that). The kept
// s_i go to a per-wave pending list and are folded (exp, log1p, compensated sums) 64 at a time: at the default cut-off a
// wave keeps ~9 of its 1024 rows per slab, a batch per slab would run 55 of 64 lanes empty. (A lane taking PAIRS of rows --
// 16-byte LDS reads and 16-byte stores of s_i, half the instructions -- measured no better: 0.885 against 0.881 ms with s_i
// written; the row output costs its 0.8 GB of writes, not its store instructions.)
template <bool ROWS, bool MASK>
__device__ __forceinline__ void bs_epilogue(double* sums, double* pend, int& n_pend, int64_t s, int wave, int lane, Acc& acc, const BsScalars& sc,
                                            const int64_t* __restrict__ row_ptr, double* __restrict__ row_logp_out,
                                            uint64_t* __restrict__ keep_mask_out) {
    constexpr int U = 8, NI = kBsSlabRows / kBsWaves / kWave; // 16 rows per lane
    const int base = wave * (kBsSlabRows / kBsWaves);
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += U) {
        double sv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) sv[u] = sums[base + (i0 + u) * kWave + lane];
#pragma unroll
        for (int u = 0; u < U; ++u) sums[base + (i0 + u) * kWave + lane] = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = s * kBsSlabRows + base + (i0 + u) * kWave + lane;
            const bool valid = r < sc.n_rows;
            int len = 0;
            if (sc.need_len && valid) len = (int)(row_ptr[r + 1] - row_ptr[r]);
            const bool keep = valid && (sv[u] >= sc.log_cut) && (sc.limit_order == 0 || len <= sc.limit_order);
            const unsigned long long m = __ballot(keep);
            if (ROWS && valid) row_logp_out[r] = sv[u];
            if (MASK && lane == 0 && r < sc.n_rows) keep_mask_out[r >> 6] = m;
            if (m != 0) { // wave-uniform: most 64-row groups have no survivor under the default cut-off
                if (keep) {
                    acc.n_kept += 1.0;
                    if (sv[u] > acc.max_logp) { acc.max_logp = sv[u]; acc.argmax = (double)(sc.row_offset + r); }
                    pend[n_pend + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = sv[u];
                }
                n_pend += __builtin_popcountll(m);
                if (n_pend >= kWave) { // a full batch: the last 64 of the list
                    n_pend -= kWave;
                    fold_kept_row(acc.rare, acc.l1p, acc.n_certain, pend[n_pend + lane], true);
                }
            }
        }
    }
    if (wave == 0) sums[kBsSlabRows + lane] = 0.0; // the sinks
}

// The stream is read with NON-TEMPORAL loads: its lines then do not push the logp table (which the gathers read through L1 / L2)
// out of the caches: 0.866 -> 0.839 ms at config 3 with every s_i written (-DBS_NT=0: ordinary loads). The passes this kernel serves
// run one at a time (per-row outputs are not pipelined), which is where such loads pay: in k_fr_filter_ls they cost 4 % when two
// passes share the memory system, and in k_fx2 (whose passes are pipelined) they bought nothing, 0.7135 against 0.709 ms.
#ifndef BS_NT
#define BS_NT 1
#endif
__device__ __forceinline__ uint4 BS_LOAD(const uint4* p) {
#if BS_NT
    typedef uint32_t bs_u32x4 __attribute__((ext_vector_type(4)));
    const bs_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const bs_u32x4*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
template <bool ROWS, bool MASK>
__global__ __launch_bounds__(kBsThreads) __attribute__((amdgpu_waves_per_eu(kBsWaves * BS_WGS / 4))) void k_bs(const uint4* __restrict__ stream, const int64_t* __restrict__ ss_off,
                                                   const int32_t* __restrict__ ss_used, const double* __restrict__ logp,
                                                   const int64_t* __restrict__ row_ptr, BsScalars sc, double* __restrict__ block_partials,
                                                   double* __restrict__ row_logp_out, uint64_t* __restrict__ keep_mask_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sums = reinterpret_cast<double*>(smem); // [16384 + 64]
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    double* pend = reinterpret_cast<double*>(smem + kBsSumsBytes) + (size_t)wave * kBsPend; // this wave's kept s_i not yet folded
    int n_pend = 0;                                                                          // (wave-uniform)
    for (int i = tid; i < kBsSlabRows + kWave; i += kBsThreads) sums[i] = 0.0;
    __syncthreads();
    Acc acc;
    acc_init(acc);
    const int64_t G = (int64_t)gridDim.x;
    // load cursor: across slab boundaries; past the workgroup's last slab: the no-op superstep (padding only)
    // (a contiguous block of slabs per workgroup, the first n_slabs % G workgroups one more: what measured 2 % faster than
    // slabs blockIdx, blockIdx + G, ... in the headline filter)
    const int64_t q_wg = sc.n_slabs / G, r_wg = sc.n_slabs % G, b_wg = (int64_t)blockIdx.x;
    const int64_t s_begin = b_wg * q_wg + (b_wg < r_wg ? b_wg : r_wg), s_end = s_begin + q_wg + (b_wg < r_wg ? 1 : 0);
    int64_t ls = s_begin, lp = 0, lp_end = 0;
    if (ls < s_end) { lp = bs_uload(ss_off + ls); lp_end = lp + __builtin_amdgcn_readfirstlane(ss_used[ls]); }
#define BS_NEXT(Q)                                                                                         \
    {                                                                                                      \
        while (lp == lp_end) {                                                                             \
            ls += 1;                                                                                       \
            if (ls < s_end) { lp = bs_uload(ss_off + ls); lp_end = lp + __builtin_amdgcn_readfirstlane(ss_used[ls]); } \
            else { lp = sc.total_ss; lp_end = sc.total_ss + 1; break; }                                     \
        }                                                                                                  \
        Q = BS_LOAD(&stream[(lp * kBsWaves + wave) * kWave + lane]);                                       \
        if (ls < s_end) ++lp;                                                                              \
    }
    int64_t s = s_begin;
    if (s < s_end) {
        // Three register sets in rotation (entries + gathered values), the loop body written out three times: a copy
        // "cur = next" at the end of a single body makes the compiler wait for EVERY load in flight each iteration (the
        // gathers just issued, the stream load behind them), i.e. one memory latency per superstep. (A fourth set of entries --
        // the stream load three supersteps ahead -- measured no better: 0.843 against 0.845 ms with s_i written.)
        uint4 e0, e1, e2;
        double v0[4], v1[4], v2[4];
        BS_NEXT(e0)
        v0[0] = logp[e0.x >> kBsRowBits]; v0[1] = logp[e0.y >> kBsRowBits]; v0[2] = logp[e0.z >> kBsRowBits]; v0[3] = logp[e0.w >> kBsRowBits];
        BS_NEXT(e1)
        int rem = __builtin_amdgcn_readfirstlane(ss_used[s]);
        // EA / VA: the superstep whose adds happen now (its gathers went out one step ago); EB: loaded, its gathers go out now
        // into VB; EC: the stream load of the superstep after that
#ifndef BS_ORDER
#define BS_ORDER 1 // (0 / 1 / 2 / 3 at config 3 with s_i written, one box: 0.877 / 0.833 / 0.858 / 0.909 ms; also measured: the
                   // load in front of the adds 0.860, the load in front of the first drain's gathers 0.885, three gathers + the load in the
                   // first drain 0.880 -- against 0.843 for order 1 on that box)
#endif
#define BS_ADD(V, J) __hip_atomic_fetch_add(&sums[r4[J]], V[J], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#define BS_SB __builtin_amdgcn_sched_barrier(0);
#if BS_GPB == 1 /* (geometry experiment: a barrier behind every add) */
#define BS_ORDER_BODY(EA, VA, EB, VB, EC) \
            BS_ADD(VA, 0) BS_SB VB[0] = logp[cB[0]]; BS_NEXT(EC) BS_SB bs_barrier(); BS_ADD(VA, 1) BS_SB VB[1] = logp[cB[1]]; BS_SB bs_barrier(); \
            BS_ADD(VA, 2) BS_SB VB[2] = logp[cB[2]]; BS_SB bs_barrier(); BS_ADD(VA, 3) BS_SB VB[3] = logp[cB[3]]; BS_SB bs_barrier();
#elif BS_ORDER == 0
#define BS_ORDER_BODY(EA, VA, EB, VB, EC) \
            BS_ADD(VA, 0) BS_ADD(VA, 1) BS_SB VB[0] = logp[cB[0]]; VB[1] = logp[cB[1]]; BS_SB bs_barrier(); \
            BS_ADD(VA, 2) BS_ADD(VA, 3) BS_SB VB[2] = logp[cB[2]]; VB[3] = logp[cB[3]]; BS_NEXT(EC) BS_SB bs_barrier();
#elif BS_ORDER == 1 /* the stream load in the first drain */
#define BS_ORDER_BODY(EA, VA, EB, VB, EC) \
            BS_ADD(VA, 0) BS_ADD(VA, 1) BS_SB VB[0] = logp[cB[0]]; VB[1] = logp[cB[1]]; BS_NEXT(EC) BS_SB bs_barrier(); \
            BS_ADD(VA, 2) BS_ADD(VA, 3) BS_SB VB[2] = logp[cB[2]]; VB[3] = logp[cB[3]]; BS_SB bs_barrier();
#elif BS_ORDER == 2 /* add, gather, add, gather */
#define BS_ORDER_BODY(EA, VA, EB, VB, EC) \
            BS_ADD(VA, 0) BS_SB VB[0] = logp[cB[0]]; BS_SB BS_ADD(VA, 1) BS_SB VB[1] = logp[cB[1]]; BS_SB bs_barrier(); \
            BS_ADD(VA, 2) BS_SB VB[2] = logp[cB[2]]; BS_SB BS_ADD(VA, 3) BS_SB VB[3] = logp[cB[3]]; BS_NEXT(EC) BS_SB bs_barrier();
#elif BS_ORDER == 3 /* all four gathers and the load in the FIRST drain */
#define BS_ORDER_BODY(EA, VA, EB, VB, EC) \
            BS_ADD(VA, 0) BS_ADD(VA, 1) BS_SB VB[0] = logp[cB[0]]; VB[1] = logp[cB[1]]; VB[2] = logp[cB[2]]; VB[3] = logp[cB[3]]; BS_NEXT(EC) BS_SB bs_barrier(); \
            BS_ADD(VA, 2) BS_ADD(VA, 3) BS_SB bs_barrier();
#endif
#define BS_STEP(EA, VA, EB, VB, EC)                                                                                      \
        {                                                                                                                \
            while (rem == 0) { /* slab finished (or without entries): every wave folds ITS 1024 rows, then the next slab */ \
                bs_barrier();                                                                                            \
                bs_epilogue<ROWS, MASK>(sums, pend, n_pend, s, wave, lane, acc, sc, row_ptr, row_logp_out, keep_mask_out); \
                bs_barrier();                                                                                            \
                s += 1;                                                                                                  \
                if (s >= s_end) break;                                                                                   \
                rem = __builtin_amdgcn_readfirstlane(ss_used[s]);                                                        \
            }                                                                                                            \
            if (s >= s_end) break;                                                                                       \
            /* The next superstep's gathers go out WHILE a band's adds drain (behind the adds, in front of the wait and the  \
               barrier): the vector-memory issue time then overlaps the LDS latency instead of delaying the adds: 0.87 against \
               0.90 ms with s_i written, same box. (All four gathers and the load in front of the adds: that 0.90; one gather, \
               one add in turn: 0.88 sum-only against 0.80.) */                                                            \
            const uint32_t r4[4] = {EA.x & kBsRowMask, EA.y & kBsRowMask, EA.z & kBsRowMask, EA.w & kBsRowMask};         \
            const uint32_t cB[4] = {EB.x >> kBsRowBits, EB.y >> kBsRowBits, EB.z >> kBsRowBits, EB.w >> kBsRowBits};     \
            BS_ORDER_BODY(EA, VA, EB, VB, EC)                                                                            \
            --rem;                                                                                                       \
        }
        for (;;) {
            BS_STEP(e0, v0, e1, v1, e2)
            BS_STEP(e1, v1, e2, v2, e0)
            BS_STEP(e2, v2, e0, v0, e1)
        }
#undef BS_STEP
#undef BS_ORDER_BODY
#undef BS_ADD
#undef BS_SB
    }
#undef BS_NEXT
    if (n_pend > 0) fold_kept_row(acc.rare, acc.l1p, acc.n_certain, lane < n_pend ? pend[lane] : 0.0, lane < n_pend); // (wave-uniform)
    // workgroup reduce of the wave records (the sums are dead now)
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);
    acc_wave_reduce(acc);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    __syncthreads();
    if (tid == 0) {
        Acc r;
        acc_load(r, s_red);
        for (int w = 1; w < kBsWaves; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(r, o);
        }
        acc_store(block_partials + (size_t)blockIdx.x * kAccDoubles, r);
    }
}

} // namespace

int bs_max_events() { return 1 << (32 - kBsRowBits); }
int bs_slab_rows() { return kBsSlabRows; }
int64_t bs_ss_bytes() { return (int64_t)kBsSsEntries * 4; }

int bs_default_grid(int n_cu, int64_t n_slabs) {
    int64_t g = n_slabs < (int64_t)n_cu * BS_WGS ? n_slabs : (int64_t)n_cu * BS_WGS;
    if (g < 1) g = 1;
    return (int)g;
}

// Builds the band stream of the model's canonical CSR. hipSuccess with *ok = false: the model does not fit the layout (a
// slab needs more than 8192 bands, the greedy ran out of its allotment, or -- accept_padding false -- more than 30 % of the
// slots would be padding: rows of hundreds of entries); the caller keeps the wave-slab stream.
hipError_t bs_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int64_t nnz, int32_t n_events,
                    int32_t max_row_len, bool accept_padding, uint32_t** d_stream_out, int64_t** d_off_out, int32_t** d_used_out,
                    int64_t* total_ss_out, int64_t* used_ss_out, int64_t* n_slabs_out, bool* ok, hipStream_t s) {
    *ok = false;
    const int64_t n_slabs = (n_rows + kBsSlabRows - 1) / kBsSlabRows;
    *n_slabs_out = n_slabs;
    if (n_events < 1 || n_events > bs_max_events()) return hipSuccess;
    if (n_slabs == 0) { // an empty model: one workgroup writes the empty record, nothing is read
        int64_t* off = nullptr;
        int32_t* used = nullptr;
        uint32_t* st = nullptr;
        hipError_t e0 = hipMalloc(&off, sizeof(int64_t));
        if (e0 == hipSuccess) e0 = hipMalloc(&used, sizeof(int32_t));
        if (e0 == hipSuccess) e0 = hipMalloc(&st, sizeof(uint32_t) * kBsSsEntries);
        if (e0 == hipSuccess) e0 = hipMemsetAsync(off, 0, sizeof(int64_t), s);
        if (e0 == hipSuccess) e0 = hipMemsetAsync(used, 0, sizeof(int32_t), s);
        if (e0 == hipSuccess) e0 = hipStreamSynchronize(s);
        if (e0 != hipSuccess) {
            (void)hipGetLastError();
            if (off) (void)hipFree(off);
            if (used) (void)hipFree(used);
            if (st) (void)hipFree(st);
            return e0;
        }
        *d_stream_out = st; *d_off_out = off; *d_used_out = used; *total_ss_out = 0; *used_ss_out = 0;
        *ok = true;
        return hipSuccess;
    }
    int64_t* d_off = nullptr;
    int32_t* d_used = nullptr;
    int* d_flags = nullptr; // [0] too many bands, [1] overflow
    uint32_t *d_tmp = nullptr, *d_stream = nullptr;
    uint16_t* d_cnt = nullptr;
    auto bail = [&](hipError_t err) {
        (void)hipGetLastError();
        if (d_off) (void)hipFree(d_off);
        if (d_used) (void)hipFree(d_used);
        if (d_flags) (void)hipFree(d_flags);
        if (d_tmp) (void)hipFree(d_tmp);
        if (d_stream) (void)hipFree(d_stream);
        if (d_cnt) (void)hipFree(d_cnt);
        return err;
    };
    hipError_t e = hipMalloc(&d_off, sizeof(int64_t) * (size_t)(n_slabs + 1));
    if (e == hipSuccess) e = hipMalloc(&d_used, sizeof(int32_t) * (size_t)n_slabs);
    if (e == hipSuccess) e = hipMalloc(&d_flags, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(d_flags, 0, 2 * sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_off, 0, sizeof(int64_t), s);
    if (e != hipSuccess) return bail(e);
    (void)max_row_len;
    hipLaunchKernelGGL(k_bs_alloc, dim3((unsigned)n_slabs), dim3(256), 0, s, d_row_ptr, n_rows, d_off + 1, d_flags);
    e = hipGetLastError();
    if (e == hipSuccess) e = scan_inclusive_i64(d_off + 1, n_slabs, s);
    int flags[2] = {0, 0};
    int64_t total_ss = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&total_ss, d_off + n_slabs, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    // a model of rows so long that the allotment dwarfs the entries is not a band-stream model
    const int64_t ideal_ss = (nnz + kBsSsEntries - 1) / kBsSsEntries + n_slabs;
    if (flags[0] || (!accept_padding && total_ss > 2 * ideal_ss + 8 * n_slabs)) { (void)bail(hipSuccess); return hipSuccess; }
    const size_t n_bands = (size_t)(total_ss + 1) * kBsBandsPerSs;
    e = hipMalloc(&d_tmp, sizeof(uint32_t) * (size_t)total_ss * kBsSsEntries + 16);
    if (e == hipSuccess) e = hipMalloc(&d_stream, sizeof(uint32_t) * (size_t)(total_ss + 1) * kBsSsEntries);
    if (e == hipSuccess) e = hipMalloc(&d_cnt, sizeof(uint16_t) * n_bands);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, sizeof(uint16_t) * n_bands, s); // (the no-op superstep: empty bands)
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL(k_bs_place, dim3((unsigned)n_slabs), dim3(256), 0, s, d_row_ptr, d_col_idx, n_rows, (uint32_t)n_events,
                       (const int64_t*)d_off, d_tmp, d_cnt, d_used, d_flags + 1);
    e = hipGetLastError();
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_bs_sort, dim3((unsigned)n_bands), dim3(256), 0, s, (const uint32_t*)d_tmp, (const uint16_t*)d_cnt, (int64_t)n_bands,
                           d_stream);
        e = hipGetLastError();
    }
    std::vector<int32_t> used((size_t)n_slabs);
    if (e == hipSuccess) e = hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(used.data(), d_used, sizeof(int32_t) * (size_t)n_slabs, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    int64_t used_ss = 0;
    for (int32_t u : used) used_ss += u;
    const int64_t dense_ss = (nnz + kBsSsEntries - 1) / kBsSsEntries;
    if (flags[1] || (!accept_padding && used_ss * 10 > dense_ss * 13 + 10 * n_slabs)) { (void)bail(hipSuccess); return hipSuccess; }
    (void)hipFree(d_tmp);
    (void)hipFree(d_cnt);
    (void)hipFree(d_flags);
    *d_stream_out = d_stream; *d_off_out = d_off; *d_used_out = d_used;
    *total_ss_out = total_ss; *used_ss_out = used_ss;
    *ok = true;
    return hipSuccess;
}

hipError_t launch_bs_kernel(const RowKernelArgs& ra, const uint32_t* stream, const int64_t* ss_off, const int32_t* ss_used, int64_t total_ss,
                            int64_t n_slabs, int need_len, int grid, int* n_records, hipStream_t s) {
    *n_records = grid;
    BsScalars sc;
    sc.total_ss = total_ss;
    sc.n_rows = ra.n_rows; sc.n_slabs = n_slabs; sc.row_offset = ra.row_offset;
    sc.log_cut = ra.log_cut; sc.limit_order = ra.limit_order; sc.need_len = need_len;
    const bool rows = ra.row_logp_out != nullptr, mask = ra.keep_mask_out != nullptr;
#define BS_LAUNCH(R, M)                                                                                                                  \
    {                                                                                                                                    \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bs<R, M>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / BS_WGS); \
        if (e != hipSuccess) return e;                                                                                                   \
        hipLaunchKernelGGL((k_bs<R, M>), dim3(grid), dim3(kBsThreads), kBsLdsBytes, s, reinterpret_cast<const uint4*>(stream), ss_off, ss_used, \
                           ra.logp, ra.row_ptr, sc, ra.block_partials, ra.row_logp_out, ra.keep_mask_out);                               \
    }
    if (rows && mask) BS_LAUNCH(true, true)
    else if (rows) BS_LAUNCH(true, false)
    else if (mask) BS_LAUNCH(false, true)
    else BS_LAUNCH(false, false)
#undef BS_LAUNCH
    return hipGetLastError();
}

} // namespace canopy

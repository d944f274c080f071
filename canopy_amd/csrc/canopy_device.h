// canopy_device.h -- device-side building blocks shared by the gfx950 kernels.
// Written for wave64 / CDNA4 only (no other back-end, no compatibility layer).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace canopy {

constexpr int kWave = 64;

// ---- counter-based generator (identical integer arithmetic to oracle/canopy_oracle.c) ----------
__host__ __device__ inline uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
enum : uint64_t { kStreamLen = 1, kStreamCol = 2, kStreamProb = 3 };
__host__ __device__ inline uint64_t stream_key(uint64_t seed, uint64_t stream) {
    return mix64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
}
__host__ __device__ inline uint64_t draw_keyed(uint64_t key, uint64_t index, uint64_t t) {
    return mix64(mix64(key ^ index) + t);
}
__host__ __device__ inline uint32_t bounded(uint64_t h, uint32_t n) {
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

// ---- compensated accumulation ---------------------------------------------------------------------
struct CSum {
    double s, c;
};

// Neumaier step, branch-free: the error term of a + x is recovered with Knuth's TwoSum, which
// equals Neumaier's two-branch form whenever no overflow occurs.
__device__ __forceinline__ void csum_add(CSum& a, double x) {
    const double t = a.s + x;
    const double bb = t - a.s;
    a.c += (a.s - (t - bb)) + (x - bb);
    a.s = t;
}
__device__ __forceinline__ void csum_merge(CSum& a, const CSum& b) {
    csum_add(a, b.s);
    a.c += b.c;
}

__device__ __forceinline__ double shfl_down_f64(double v, int delta) {
    return __shfl_down(v, delta, kWave);
}

// One shard-partial in registers.
struct Acc {
    CSum rare, l1p;
    double n_kept, n_certain;
    double max_logp;
    double argmax; // global row index as a double (exact < 2^53), -1 = none
};

__device__ __forceinline__ void acc_init(Acc& a) {
    a.rare = {0.0, 0.0};
    a.l1p = {0.0, 0.0};
    a.n_kept = 0.0;
    a.n_certain = 0.0;
    a.max_logp = -__builtin_inf();
    a.argmax = -1.0;
}

// a <- a (+) b where every row of a precedes... not assumed: ties resolve to the lower index.
__device__ __forceinline__ void acc_merge(Acc& a, const Acc& b) {
    csum_merge(a.rare, b.rare);
    csum_merge(a.l1p, b.l1p);
    a.n_kept += b.n_kept;
    a.n_certain += b.n_certain;
    const bool take = (b.argmax >= 0.0) &&
                      ((b.max_logp > a.max_logp) ||
                       (b.max_logp == a.max_logp && (a.argmax < 0.0 || b.argmax < a.argmax)));
    a.max_logp = take ? b.max_logp : a.max_logp;
    a.argmax = take ? b.argmax : a.argmax;
}

__device__ __forceinline__ Acc acc_shfl_down(const Acc& a, int delta) {
    Acc r;
    r.rare.s = shfl_down_f64(a.rare.s, delta);
    r.rare.c = shfl_down_f64(a.rare.c, delta);
    r.l1p.s = shfl_down_f64(a.l1p.s, delta);
    r.l1p.c = shfl_down_f64(a.l1p.c, delta);
    r.n_kept = shfl_down_f64(a.n_kept, delta);
    r.n_certain = shfl_down_f64(a.n_certain, delta);
    r.max_logp = shfl_down_f64(a.max_logp, delta);
    r.argmax = shfl_down_f64(a.argmax, delta);
    return r;
}

// wave64 tree: lane 0 ends with the wave's merged record (fixed order -> reproducible).
__device__ __forceinline__ void acc_wave_reduce(Acc& a) {
#pragma unroll
    for (int d = kWave / 2; d >= 1; d >>= 1) {
        Acc o = acc_shfl_down(a, d);
        acc_merge(a, o);
    }
}

constexpr int kAccDoubles = 8; // rare.s rare.c l1p.s l1p.c n_kept n_certain max_logp argmax

__device__ __forceinline__ void acc_store(double* dst, const Acc& a) {
    dst[0] = a.rare.s; dst[1] = a.rare.c; dst[2] = a.l1p.s; dst[3] = a.l1p.c;
    dst[4] = a.n_kept; dst[5] = a.n_certain; dst[6] = a.max_logp; dst[7] = a.argmax;
}
__device__ __forceinline__ void acc_load(Acc& a, const double* src) {
    a.rare.s = src[0]; a.rare.c = src[1]; a.l1p.s = src[2]; a.l1p.c = src[3];
    a.n_kept = src[4]; a.n_certain = src[5]; a.max_logp = src[6]; a.argmax = src[7];
}

// Contribution of one kept row to the two compensated sums: P = exp(s), log1p(-P).
// For P < 2^-12 (every realistic rare-event cut set) log1p(-P) = -P(1 + P/2 + P^2/3 + P^3/4 + P^4/5 + P^5/6)
// is exact to < 1e-19 relative; ocml's log1p runs only when some lane of the wave needs it.
__device__ __forceinline__ double log1p_neg_small(double P) {
    return -P * (1.0 + P * (0.5 + P * (1.0 / 3.0 + P * (0.25 + P * (0.2 + P * (1.0 / 6.0))))));
}
__device__ __forceinline__ void fold_terms(CSum& rare, CSum& l1p, double& n_certain, double P, double l, bool on) {
    if (on) {
        csum_add(rare, P);
        if (P < 1.0) csum_add(l1p, l); else n_certain += 1.0;
    }
}
__device__ __forceinline__ void fold_kept_row(CSum& rare, CSum& l1p, double& n_certain, double s, bool on) {
    const double P = on ? exp(s) : 0.0;
    double l = log1p_neg_small(P);
    const bool big = on && P >= 0x1.0p-12;
    if (__any(big)) {
        if (big) l = (P < 1.0) ? log1p(-P) : 0.0;
    }
    fold_terms(rare, l1p, n_certain, P, l, on);
}
// N rows per lane at once: the N exp / polynomial chains are independent, so a lone wave on a SIMD overlaps
// their latencies; the compensated adds then run in index order (same result as N fold_kept_row calls).
template <int N>
__device__ __forceinline__ void fold_kept_rows(CSum& rare, CSum& l1p, double& n_certain, const double (&s)[N],
                                               const bool (&on)[N]) {
    double P[N], l[N];
    bool any_big = false;
#pragma unroll
    for (int u = 0; u < N; ++u) P[u] = on[u] ? exp(s[u]) : 0.0;
#pragma unroll
    for (int u = 0; u < N; ++u) {
        l[u] = log1p_neg_small(P[u]);
        any_big = any_big || (on[u] && P[u] >= 0x1.0p-12);
    }
    if (__any(any_big)) {
#pragma unroll
        for (int u = 0; u < N; ++u)
            if (on[u] && P[u] >= 0x1.0p-12) l[u] = (P[u] < 1.0) ? log1p(-P[u]) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < N; ++u) fold_terms(rare, l1p, n_certain, P[u], l[u], on[u]);
}

// Deterministic final reduce of n per-workgroup records by ONE 256-thread workgroup: fixed thread <- record
// assignment, fixed tree. Output is the canopy_partial layout (hi/lo renormalised); with a gather buffer the
// record also goes into slot `rank` and zeros into the other slots (K5: one sum all-reduce = all-gather).
// s_red: 4 * kAccDoubles doubles, s_out: kAccDoubles doubles of LDS.
__device__ __forceinline__ void finalize_records(const double* __restrict__ parts, int n, double* out, double* gather,
                                                 int world, int rank, double* s_red, double* s_out) {
    Acc acc;
    acc_init(acc);
    for (int i = threadIdx.x; i < n; i += 256) {
        Acc o;
        acc_load(o, parts + (size_t)i * kAccDoubles);
        acc_merge(acc, o);
    }
    acc_wave_reduce(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    __syncthreads();
    if (threadIdx.x == 0) {
        Acc t;
        acc_load(t, s_red);
        for (int w = 1; w < 4; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(t, o);
        }
        const double rh = t.rare.s + t.rare.c;
        const double rl = (t.rare.s - rh) + t.rare.c;
        const double lh = t.l1p.s + t.l1p.c;
        const double ll = (t.l1p.s - lh) + t.l1p.c;
        s_out[0] = rh; s_out[1] = rl; s_out[2] = lh; s_out[3] = ll;
        s_out[4] = t.n_kept; s_out[5] = t.n_certain; s_out[6] = t.max_logp; s_out[7] = t.argmax;
    }
    __syncthreads();
    if (threadIdx.x < kAccDoubles) out[threadIdx.x] = s_out[threadIdx.x];
    if (gather != nullptr)
        for (int i = threadIdx.x; i < world * kAccDoubles; i += 256)
            gather[i] = (i / kAccDoubles == rank) ? s_out[i % kAccDoubles] : 0.0;
}

// Per-row epilogue shared by every row kernel: filter, P = exp(s), both sums, argmax.
__device__ __forceinline__ bool row_epilogue(Acc& a, double s, int len, bool valid, double log_cut,
                                            int limit_order, double global_row) {
    const bool keep = valid && (s >= log_cut) && (limit_order == 0 || len <= limit_order);
    fold_kept_row(a.rare, a.l1p, a.n_certain, s, keep);
    if (keep) {
        a.n_kept += 1.0;
        if (s > a.max_logp) { a.max_logp = s; a.argmax = global_row; }
    }
    return keep;
}

// Fold the RW finished row sums of one wave-slab and clear them. Two passes:
//   1. per row: filter decision, count, argmax, optional outputs (cheap, every lane busy); the kept
//      s_i are compacted to the front of the wave's own sums slice;
//   2. exp / log1p / compensated sums over the compacted list only, 64 kept rows per instruction.
// (With the default cut-off < 1 % of the rows survive: the transcendental work all but vanishes.)
template <int RW>
__device__ __noinline__ void ws_epilogue(double* sums, int64_t t, int lane, Acc* acc_io, int64_t n_rows,
                                         int64_t row_offset, double log_cut, int limit_order, int need_len,
                                         const int64_t* row_ptr, double* row_logp_out, uint64_t* keep_mask_out,
                                         int it_begin, int it_end) {
    // rows [it_begin*64, it_end*64) of the wave-slab are this wave's share; it compacts into the front
    // of its own share only
    Acc acc = *acc_io;
    const int comp_base = it_begin * kWave;
    int n_comp = 0; // wave-uniform
    constexpr int U = 8;
    for (int i0 = it_begin; i0 < it_end; i0 += U) {
        double s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = sums[(i0 + u) * kWave + lane];
#pragma unroll
        for (int u = 0; u < U; ++u) sums[(i0 + u) * kWave + lane] = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t r = t * RW + (i0 + u) * kWave + lane;
            const bool valid = r < n_rows;
            int len = 0;
            if (need_len && valid) len = (int)(row_ptr[r + 1] - row_ptr[r]);
            const bool keep = valid && (s[u] >= log_cut) && (limit_order == 0 || len <= limit_order);
            const unsigned long long m = __ballot(keep);
            if (m != 0) { // wave-uniform: most 64-row groups have no survivor under the default cut-off
                if (keep) {
                    acc.n_kept += 1.0;
                    if (s[u] > acc.max_logp) { acc.max_logp = s[u]; acc.argmax = (double)(row_offset + r); }
                    const int pos = comp_base + n_comp +
                                    (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    sums[pos] = s[u]; // pos <= this row's own slot, already read and cleared above
                }
                n_comp += __builtin_popcountll(m);
            }
            if (row_logp_out != nullptr && valid) row_logp_out[r] = s[u];
            if (keep_mask_out != nullptr && lane == 0 && r < n_rows) keep_mask_out[r >> 6] = m;
        }
    }
    int j = 0;
    for (; j + kWave < n_comp; j += 4 * kWave) { // more than one 64-row batch left: four independent exp chains
        double sv[4];
        bool on[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            on[u] = j + u * kWave + lane < n_comp;
            sv[u] = on[u] ? sums[comp_base + j + u * kWave + lane] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (on[u]) sums[comp_base + j + u * kWave + lane] = 0.0;
        fold_kept_rows<4>(acc.rare, acc.l1p, acc.n_certain, sv, on);
    }
    for (; j < n_comp; j += kWave) {
        const bool on = j + lane < n_comp;
        const double sv = on ? sums[comp_base + j + lane] : 0.0;
        if (on) sums[comp_base + j + lane] = 0.0;
        fold_kept_row(acc.rare, acc.l1p, acc.n_certain, sv, on);
    }
    if (it_begin == 0) sums[RW + lane] = 0.0; // the sink
    *acc_io = acc;
}

} // namespace canopy

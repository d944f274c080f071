// kernels_ws.hip -- the "wave-slab stream" (WS) variant of K1..K4 for large basic-event counts,
// gfx950 / wave64 only.
//
// Why: with E = 1e5 the logp table (0.8 MB) cannot live in LDS, and random 8-byte gathers through
// L1/L2 run at the L2 request rate (~0.28e12/s measured, profiles/microbench_r01.txt), 5x below what
// the HBM stream needs. So the model upload derives a second layout in which the gathers are
// column-local and the per-row order of additions is still exactly the CSR order:
//
//   * rows are cut into wave-slabs of RW consecutive rows; ONE wave owns a wave-slab: the running
//     sums s_i of its RW rows live in that wave's private LDS slice (no other wave touches them);
//   * the wave-slab's entries are re-ordered by column into groups of 64 packed u32
//     (col << 13 | row_local), every group holding 64 DISTINCT rows, and a row's entries appear in
//     strictly increasing group order (= ascending column = CSR order);
//   * one wave instruction handles one group: 64 gathers logp[col] that fall into a narrow column
//     window (few cache lines), then 64 LDS fp64 adds to distinct slots. LDS operations of one wave
//     execute in issue order, so row i receives logp[c0], logp[c1], ... in CSR order: s_i is the same
//     chain of IEEE additions as in the ROWS kernels and in the oracle -> bit-identical.
//   * no barriers, no cross-wave traffic; the stream is read with 16-byte coalesced loads through a
//     register ring that runs ahead across wave-slab boundaries.
//
// Stream layout: chunk = 256 entries = 4 groups; group g of a wave-slab is chunk g/4, and its
// slot k (0..63) sits at entry index 4*k + (g%4) of the chunk, so lane k's 16-byte load brings its
// entry of each of the chunk's 4 groups. Unused slots hold the no-op entry (col 0, row_local RW).
#include <stdlib.h>

#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

constexpr int kRowBits = 13;
constexpr uint32_t kRowMask = (1u << kRowBits) - 1;
constexpr int kGroup = 64;
constexpr int kChunk = 256; // entries per chunk (one uint4 per lane)

template <int RW>
struct WsCfg {
    static_assert(RW <= 8192 && (RW % 64) == 0, "row_local needs RW < 2^(row_bits) with the sink slot");
    static constexpr int row_bits = RW > 4096 ? 14 : 13;
    static constexpr int waves = 16384 / RW;     // per workgroup = per CU
    static constexpr int slots = RW + 64;        // sink at RW (.. RW+63 unused padding)
    static constexpr size_t lds_bytes = (size_t)waves * slots * 8;
};

// ---- builder: greedy group assignment, one thread per wave-slab (sequential => deterministic) ----
//
// target group of an entry = its column's share of the wave-slab's nominal group count; it is
// pushed later while (a) the row's previous entry sits in the same or a later group, (b) the group
// is full. Pass WRITE=false only counts the groups used.
template <bool WRITE>
__global__ __launch_bounds__(64) void k_ws_build(const int64_t* __restrict__ row_ptr,
                                                 const int32_t* __restrict__ col_idx, int64_t n_rows,
                                                 int32_t n_events, int rw, int row_bits, int64_t n_ws, int gcap,
                                                 uint8_t* cnt_scratch, int64_t* chunks /* [n_ws] out (count pass) */,
                                                 const int64_t* __restrict__ ws_off, uint32_t* stream,
                                                 int* overflow) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_ws) return;
    const int64_t r0 = t * rw;
    const int64_t r1 = (r0 + rw < n_rows) ? r0 + rw : n_rows;
    const int64_t nnz = row_ptr[r1] - row_ptr[r0];
    int64_t ng_nom = (nnz + kGroup - 1) / kGroup;
    if (ng_nom < 1) ng_nom = 1;
    uint8_t* cnt = cnt_scratch + (size_t)t * gcap;
    for (int g = 0; g < gcap; ++g) cnt[g] = 0;
    int g_max = -1;
    const int64_t base = WRITE ? ws_off[t] * kChunk : 0;
    bool ovf = false;
    for (int64_t r = r0; r < r1 && !ovf; ++r) {
        int g_prev = -1;
        const int64_t b = row_ptr[r], e = row_ptr[r + 1];
        for (int64_t k = b; k < e; ++k) {
            const uint32_t c = (uint32_t)col_idx[k];
            int g = (int)(((uint64_t)c * (uint64_t)ng_nom) / (uint64_t)n_events);
            if (g <= g_prev) g = g_prev + 1;
            while (g < gcap && cnt[g] >= kGroup) ++g;
            if (g >= gcap) { ovf = true; break; }
            const int slot = cnt[g]++;
            if (WRITE) stream[base + (int64_t)(g >> 2) * kChunk + slot * 4 + (g & 3)] = (c << row_bits) | (uint32_t)(r - r0);
            g_prev = g;
            g_max = g > g_max ? g : g_max;
        }
    }
    if (ovf) atomicExch(overflow, 1);
    if (!WRITE) chunks[t] = (int64_t)((g_max + 1 + 3) / 4);
}

__global__ __launch_bounds__(256) void k_fill_u32(uint32_t* p, int64_t n, uint32_t v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // no-op entries: col 0 -> the 64 sink slots, slot k of every group on its own (64 LDS adds on ONE address serialise)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v + (uint32_t)((i >> 2) & 63);
}

// ---- the stream kernel ---------------------------------------------------------------------------

struct WsScalars {
    int64_t total_chunks; // ws_off[n_ws]; the stream has >= 1 no-op chunk after it
    int64_t n_rows, n_ws, row_offset;
    double log_cut;
    int32_t limit_order;
    int32_t need_len;
};

// wave-uniform 64-bit load: the address is uniform, the value is forced into SGPRs
__device__ __forceinline__ int64_t uload_i64(const int64_t* __restrict__ p) {
    const int64_t v = *p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | (uint64_t)lo);
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

template <int RW, int D, int ABL>
__global__ __launch_bounds__(WsCfg<RW>::waves * kWave) void k_ws(const uint4* __restrict__ stream,
                                                                 const int64_t* __restrict__ ws_off,
                                                                 const double* __restrict__ logp,
                                                                 const int64_t* __restrict__ row_ptr, WsScalars sc,
                                                                 double* __restrict__ block_partials,
                                                                 double* __restrict__ row_logp_out,
                                                                 uint64_t* __restrict__ keep_mask_out) {
    using C = WsCfg<RW>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    double* sums = reinterpret_cast<double*>(smem) + (size_t)wave * C::slots;
    for (int i = lane; i < C::slots; i += kWave) sums[i] = 0.0;

    Acc acc;
    acc_init(acc);

    const int64_t GW = (int64_t)gridDim.x * C::waves;
    int64_t t = (int64_t)blockIdx.x * C::waves + wave; // wave-uniform (SGPRs)

    if (t < sc.n_ws) {
        // Per batch of D chunks: (G) issue ALL 4*D gathers of the batch, (L) issue the D stream loads
        // of the NEXT batch into the other register set, (A) the LDS adds. vmcnt retires in order:
        // the adds wait only for gathers, which are OLDER than the stream loads just issued, so the
        // HBM loads stay in flight under the whole add phase (and under the other waves' work).
        int64_t lt = t;
        int64_t lp = uload_i64(ws_off + lt), lp_end = uload_i64(ws_off + lt + 1);
        int64_t rem = lp_end - lp;
        uint4 q[2][D];
        double v[D][4];
        bool done = false;
#define WS_LOAD_BATCH(QB)                                                                              \
        _Pragma("unroll") for (int k = 0; k < D; ++k) {                                                \
            while (lp == lp_end) { /* skip wave-slabs without entries; past the end: the no-op chunk */ \
                lt += GW;                                                                              \
                if (lt < sc.n_ws) { lp = uload_i64(ws_off + lt); lp_end = uload_i64(ws_off + lt + 1); } \
                else { lp = sc.total_chunks; lp_end = sc.total_chunks + 1; break; }                     \
            }                                                                                          \
            QB[k] = stream[lp * kWave + lane];                                                         \
            ++lp;                                                                                      \
        }
#define WS_BATCH(QC, QN)                                                                               \
        if (!done) {                                                                                   \
            _Pragma("unroll") for (int k = 0; k < D; ++k) {                                            \
                const uint32_t e[4] = {QC[k].x, QC[k].y, QC[k].z, QC[k].w};                            \
                _Pragma("unroll") for (int u = 0; u < 4; ++u)                                          \
                    v[k][u] = (ABL & 1) ? (double)e[u] : logp[e[u] >> C::row_bits];                       \
            }                                                                                          \
            WS_LOAD_BATCH(QN)                                                                          \
            _Pragma("unroll") for (int k = 0; k < D; ++k) {                                            \
                if (!done) {                                                                           \
                    while (rem == 0) { /* wave-slab finished (or empty): fold its rows, move on */     \
                        if (!(ABL & 4))                                                                \
                            ws_epilogue<RW>(sums, t, lane, &acc, sc.n_rows, sc.row_offset, sc.log_cut, \
                                            sc.limit_order, sc.need_len, row_ptr, row_logp_out, keep_mask_out, 0, RW / kWave); \
                        t += GW;                                                                       \
                        if (t >= sc.n_ws) { done = true; break; }                                      \
                        rem = uload_i64(ws_off + t + 1) - uload_i64(ws_off + t);                       \
                    }                                                                                  \
                }                                                                                      \
                if (!done) {                                                                           \
                    const uint32_t e[4] = {QC[k].x, QC[k].y, QC[k].z, QC[k].w};                        \
                    double* dst[4]; /* all four slot addresses first: keeps dependent issues apart */   \
                    _Pragma("unroll") for (int u = 0; u < 4; ++u) dst[u] = &sums[e[u] & ((1u << C::row_bits) - 1)]; \
                    __builtin_amdgcn_sched_barrier(0);                                                 \
                    _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                    \
                        if (ABL & 2) asm volatile("" ::"v"(v[k][u]), "v"(e[u]));                       \
                        else                                                                           \
                            __hip_atomic_fetch_add(dst[u], v[k][u], __ATOMIC_RELAXED,                   \
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);                      \
                    }                                                                                  \
                    --rem;                                                                             \
                }                                                                                      \
            }                                                                                          \
        }
        WS_LOAD_BATCH(q[0])
        while (!done) {
            WS_BATCH(q[0], q[1])
            WS_BATCH(q[1], q[0])
        }
#undef WS_BATCH
#undef WS_LOAD_BATCH
    }

    // workgroup reduce of the wave records (the sums slices are dead now)
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);
    acc_wave_reduce(acc);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    __syncthreads();
    if (tid == 0) {
        Acc r;
        acc_load(r, s_red);
        for (int w = 1; w < C::waves; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(r, o);
        }
        acc_store(block_partials + (size_t)blockIdx.x * kAccDoubles, r);
    }
}

// Post-pass of the builder: the 64 slots of every group re-ordered by column (a group holds 64 DISTINCT rows, so any
// order inside it is the same arithmetic). One gather instruction of k_ws = one group: with ascending lane addresses,
// neighbouring lanes in the same cache line, the texture addresser takes it in ~30 clocks instead of ~41
// (benchmarks/microbench5.hip). One wave per chunk: a 64-lane bitonic sort per group (component of the lane's uint4).
__global__ __launch_bounds__(256) void k_ws_sort_groups(uint4* __restrict__ stream, int64_t n_chunks) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t c = (int64_t)blockIdx.x * (256 / kWave) + (threadIdx.x / kWave);
    if (c >= n_chunks) return; // wave-uniform
    uint4 q = stream[c * kWave + lane];
    uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 2; k <= kWave; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool take_min = ((lane & j) == 0) == ((lane & k) == 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t o = (uint32_t)__shfl_xor((int)v[u], j, kWave);
                v[u] = take_min ? (v[u] < o ? v[u] : o) : (v[u] > o ? v[u] : o);
            }
        }
    }
    stream[c * kWave + lane] = make_uint4(v[0], v[1], v[2], v[3]);
}

int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

int ws_rows_per_wave() { return 4096; }
int ws_max_events() { return 1 << (32 - (ws_rows_per_wave() > 4096 ? 14 : 13)); }

int ws_default_grid(int n_cu, int64_t n_ws) {
    const int waves = 16384 / ws_rows_per_wave();
    int64_t g = (n_ws + waves - 1) / waves;
    if (g > n_cu) g = n_cu;
    if (g < 1) g = 1;
    return (int)g;
}

// Builds the stream for the model's canonical CSR. Returns hipSuccess with *ok = false when the
// greedy assignment did not fit its group budget (the caller then keeps using the ROWS kernel).
hipError_t ws_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int32_t n_events,
                    int32_t max_row_len, uint32_t** d_stream_out, int64_t** d_ws_off_out,
                    int64_t* total_chunks_out, int64_t* n_ws_out, bool* ok, hipStream_t s) {
    *ok = false;
    const int rw = ws_rows_per_wave();
    const int64_t n_ws = (n_rows + rw - 1) / rw;
    *n_ws_out = n_ws;
    if (n_ws == 0) {
        int64_t* off = nullptr;
        uint32_t* st = nullptr;
        hipError_t e = hipMalloc(&off, sizeof(int64_t));
        if (e != hipSuccess) return e;
        e = hipMalloc(&st, sizeof(uint32_t) * kChunk);
        if (e != hipSuccess) { (void)hipFree(off); return e; }
        (void)hipMemsetAsync(off, 0, sizeof(int64_t), s);
        hipLaunchKernelGGL(k_fill_u32, dim3(1), dim3(256), 0, s, st, (int64_t)kChunk, (uint32_t)rw);
        *d_stream_out = st; *d_ws_off_out = off; *total_chunks_out = 0; *ok = true;
        return hipStreamSynchronize(s);
    }
    const int64_t max_nnz_ws = (int64_t)rw * (max_row_len > 0 ? max_row_len : 1);
    int64_t gcap64 = (max_nnz_ws + kGroup - 1) / kGroup;
    gcap64 = gcap64 + gcap64 / 4 + max_row_len + 16;
    if (gcap64 > (1 << 20)) return hipSuccess; // absurd rows: not a WS model
    const int gcap = (int)gcap64;
    uint8_t* d_cnt = nullptr;
    int64_t* d_off = nullptr;
    int* d_ovf = nullptr;
    uint32_t* d_stream = nullptr;
    hipError_t e = hipMalloc(&d_cnt, (size_t)n_ws * gcap);
    if (e == hipSuccess) e = hipMalloc(&d_off, sizeof(int64_t) * (size_t)(n_ws + 1));
    if (e == hipSuccess) e = hipMalloc(&d_ovf, sizeof(int));
    auto bail = [&](hipError_t err) {
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_off) (void)hipFree(d_off);
        if (d_ovf) (void)hipFree(d_ovf);
        if (d_stream) (void)hipFree(d_stream);
        return err;
    };
    if (e != hipSuccess) return bail(e);
    e = hipMemsetAsync(d_ovf, 0, sizeof(int), s);
    if (e != hipSuccess) return bail(e);
    // count pass: chunks per wave-slab into d_off[1..n_ws], then inclusive scan => offsets
    e = hipMemsetAsync(d_off, 0, sizeof(int64_t), s);
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL((k_ws_build<false>), dim3(grid_for(n_ws, 64)), dim3(64), 0, s, d_row_ptr, d_col_idx, n_rows,
                       n_events, rw, rw > 4096 ? 14 : 13, n_ws, gcap, d_cnt, d_off + 1, (const int64_t*)nullptr, (uint32_t*)nullptr, d_ovf);
    e = hipGetLastError();
    if (e == hipSuccess) e = scan_inclusive_i64(d_off + 1, n_ws, s);
    int ovf = 0;
    int64_t total = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&ovf, d_ovf, sizeof(int), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&total, d_off + n_ws, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    if (ovf) { (void)bail(hipSuccess); return hipSuccess; }
    const int64_t n_entries = (total + 1) * kChunk; // + one no-op chunk the ring may run into
    e = hipMalloc(&d_stream, sizeof(uint32_t) * (size_t)n_entries);
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL(k_fill_u32, dim3(grid_for(n_entries, 256) > 4096 ? 4096 : grid_for(n_entries, 256)), dim3(256), 0, s,
                       d_stream, n_entries, (uint32_t)rw);
    hipLaunchKernelGGL((k_ws_build<true>), dim3(grid_for(n_ws, 64)), dim3(64), 0, s, d_row_ptr, d_col_idx, n_rows,
                       n_events, rw, rw > 4096 ? 14 : 13, n_ws, gcap, d_cnt, (int64_t*)nullptr, (const int64_t*)d_off, d_stream, d_ovf);
    e = hipGetLastError();
    static const bool no_sort = CANOPY_TUNE_ENV("CANOPY_WS_NOSORT") != nullptr; // tuning only
    if (e == hipSuccess && total > 0 && !no_sort) {
        hipLaunchKernelGGL(k_ws_sort_groups, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, reinterpret_cast<uint4*>(d_stream), total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    (void)hipFree(d_cnt);
    (void)hipFree(d_ovf);
    *d_stream_out = d_stream;
    *d_ws_off_out = d_off;
    *total_chunks_out = total;
    *ok = true;
    return hipSuccess;
}

namespace {
template <int RW, int D, int ABL>
hipError_t launch_ws_t(const RowKernelArgs& ra, const uint32_t* stream, const int64_t* ws_off, const WsScalars& sc,
                       int grid, hipStream_t s) {
    using C = WsCfg<RW>;
    // per device, cheap: every launch
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ws<RW, D, ABL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_ws<RW, D, ABL>), dim3(grid), dim3(C::waves * kWave), C::lds_bytes, s,
                       reinterpret_cast<const uint4*>(stream), ws_off, ra.logp, ra.row_ptr, sc, ra.block_partials,
                       ra.row_logp_out, ra.keep_mask_out);
    return hipGetLastError();
}
} // namespace

hipError_t launch_ws_kernel(const RowKernelArgs& ra, const uint32_t* stream, const int64_t* ws_off,
                            int64_t total_chunks, int64_t n_ws, int need_len, int grid, int* n_records,
                            hipStream_t s) {
    *n_records = grid;
    WsScalars sc;
    sc.total_chunks = total_chunks;
    sc.n_rows = ra.n_rows; sc.n_ws = n_ws; sc.row_offset = ra.row_offset;
    sc.log_cut = ra.log_cut; sc.limit_order = ra.limit_order; sc.need_len = need_len;
#ifdef CANOPY_ABLATIONS // benchmarks/ experiments only (the timing ablations produce wrong results)
    static const int dbg = CANOPY_TUNE_ENV("CANOPY_WS_DEBUG") ? atoi(CANOPY_TUNE_ENV("CANOPY_WS_DEBUG")) : 0;
    switch (dbg) {
        case 1: return launch_ws_t<4096, 8, 1>(ra, stream, ws_off, sc, grid, s); // no gathers
        case 2: return launch_ws_t<4096, 8, 2>(ra, stream, ws_off, sc, grid, s); // no LDS adds
        case 4: return launch_ws_t<4096, 8, 4>(ra, stream, ws_off, sc, grid, s); // no epilogue
        case 7: return launch_ws_t<4096, 8, 7>(ra, stream, ws_off, sc, grid, s); // stream loads only
        case 10: return launch_ws_t<4096, 10, 0>(ra, stream, ws_off, sc, grid, s);
        case 12: return launch_ws_t<4096, 12, 0>(ra, stream, ws_off, sc, grid, s);
        case 6: return launch_ws_t<4096, 6, 0>(ra, stream, ws_off, sc, grid, s);
        default: break;
    }
#endif
    return launch_ws_t<4096, 8, 0>(ra, stream, ws_off, sc, grid, s);
}


// =====================================================================================================================
// PART_SWEEP (round 3): the same wave-owned row sums, with the table gathered from LDS instead of through the texture path.
//
// What bounds every exact pass that gathers the fp64 table through L1/L2 is the texture addresser: a 64-lane gather costs
// it ~2.2 clocks per 128-byte line touched (profiles/r03_microbench_gather_floor.txt: 29 clocks at 9 lines per instruction,
// 60 at 25, 109 at 47), and the fewer rows a wave or workgroup owns the sparser its column-sorted entries are. An LDS
// gather costs 13-16 clocks whatever the addresses. So: the columns are cut into PARTS of 8192 (64 KB of table), a
// workgroup of 8 waves owns a slab-set of 8 x 1024 rows, and for every part in turn it (1) puts the part of the table into
// LDS with coalesced loads (the L2 feeds the 0.8 MB table at > 25 TB/s), (2) every wave adds the entries of ITS 1024 rows
// that fall into the part -- ds_read_b64 of the value, ds_add_f64 into the row's sum in the wave's private LDS slice.
// Parts ascend, columns ascend inside a row, a wave's LDS operations execute in issue order, the 64 entries of an
// instruction hit 64 distinct rows: row i receives exactly ((logp[c0] + logp[c1]) + ...) in CSR order -- bit-exact per-row
// sums with NO gather through the texture path and no ordering constraint between waves.
//
// Stream ("part stream"): for every (wave-slab, part) a run of groups of 64 packed entries
//   u32 = col_in_part (14 bits: 0..8191, 8192 = the table's zero slot) << 11 | row_local (11 bits: 0..1023, 1024.. = sinks)
// a group holds 64 distinct rows, a row's entries sit in strictly increasing groups; the runs of a wave-slab follow each
// other part by part; goff[ws * H + h] = first group of the run. Padding = "zero slot -> sink row lane k".
// =====================================================================================================================
namespace {

constexpr int kPsRows = 1024;   // rows per wave
constexpr int kPsWaves = 8;     // waves per workgroup: a slab-set of 8192 rows, one workgroup per CU
constexpr int kPsPartBits = 13; // 8192 columns per part
constexpr int kPsPart = 1 << kPsPartBits;
constexpr int kPsRowBits = 11;
constexpr int kPsSlots = kPsRows + 64;
constexpr int kPsGP = 10;       // groups of one (wave-slab, part) that travel in registers (the rest is loaded when needed)
constexpr size_t kPsTabBytes = (size_t)(kPsPart + 2) * 8;
constexpr size_t kPsLdsBytes = kPsTabBytes + (size_t)kPsWaves * kPsSlots * 8;
constexpr int kPsMaxParts = 63; // the offsets of a wave-slab travel in one VGPR, lane h = part h

__device__ __host__ constexpr uint32_t ps_pad_entry(int lane) { return ((uint32_t)kPsPart << kPsRowBits) | (uint32_t)(kPsRows + lane); }

// builder: one thread per (wave-slab, part), greedy and sequential => deterministic. WRITE = false counts the groups.
template <bool WRITE>
__global__ __launch_bounds__(64) void k_ps_build(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                 int64_t n_rows, int H, int64_t n_tasks, int gcap, uint8_t* cnt_scratch,
                                                 int64_t* ng /* [n_tasks] out (count pass) */, const int64_t* __restrict__ goff,
                                                 uint32_t* stream, int* overflow) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tasks) return;
    const int64_t ws = t / H;
    const int h = (int)(t - ws * H);
    const int64_t r0 = ws * kPsRows;
    const int64_t r1 = r0 + kPsRows < n_rows ? r0 + kPsRows : n_rows;
    uint8_t* cnt = cnt_scratch + (size_t)t * gcap;
    for (int g = 0; g < gcap; ++g) cnt[g] = 0;
    int lo = 0, g_max = -1;
    const int64_t base = WRITE ? goff[t] * kGroup : 0;
    bool ovf = false;
    // Two walks: first the rows with SEVERAL entries in this part (each needs a chain of increasing groups: placed while the
    // groups are empty, the chains start at the front), then the rows with one entry, which fill every hole from the lowest
    // group up -- the run ends up with ceil(entries / 64) groups unless a chain is longer than that.
    for (int pass = 0; pass < 2 && !ovf; ++pass) {
        for (int64_t r = r0; r < r1 && !ovf; ++r) {
            const int64_t b = row_ptr[r], e = row_ptr[r + 1];
            int64_t kb = e, ke = e; // the row's entries in part h: [kb, ke)
            for (int64_t k = b; k < e; ++k) {
                const int part = (int)((uint32_t)col_idx[k] >> kPsPartBits);
                if (part < h) continue;
                if (part == h && kb == e) kb = k;
                if (part > h) { ke = k; break; } // columns ascend inside a row
            }
            if (kb == e) continue;
            const bool chain = ke - kb > 1;
            if (chain != (pass == 0)) continue;
            int g_prev = -1;
            for (int64_t k = kb; k < ke; ++k) {
                const uint32_t c = (uint32_t)col_idx[k];
                int g = g_prev + 1 > lo ? g_prev + 1 : lo;
                while (g < gcap && cnt[g] >= kGroup) ++g;
                if (g >= gcap) { ovf = true; break; }
                const int slot = cnt[g]++;
                if (WRITE) stream[base + (int64_t)g * kGroup + slot] = ((c & (uint32_t)(kPsPart - 1)) << kPsRowBits) | (uint32_t)(r - r0);
                g_prev = g;
                g_max = g > g_max ? g : g_max;
                while (lo < gcap && cnt[lo] >= kGroup) ++lo;
            }
        }
    }
    if (ovf) atomicExch(overflow, 1);
    if (!WRITE) ng[t] = (int64_t)(g_max + 1);
}

__global__ __launch_bounds__(256) void k_ps_fill(uint32_t* p, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = ps_pad_entry((int)(i & 63));
}

__global__ __launch_bounds__(256) void k_ps_narrow(const int64_t* __restrict__ src, uint32_t* __restrict__ dst, int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = (uint32_t)src[i];
}

struct PsScalars {
    int64_t n_rows, n_ws, row_offset;
    double log_cut;
    int32_t limit_order, need_len, H, n_events;
};

__device__ __forceinline__ uint32_t ps_lane_u32(uint32_t v, int lane_idx) { // v's value in lane `lane_idx` (wave-uniform index)
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane_idx);
}

// one group of 64 entries: value out of the LDS table, add into the owner row's LDS sum
#define PS_GROUP4(E0, E1, E2, E3)                                                                                         \
    do {                                                                                                                  \
        const uint32_t e_[4] = {E0, E1, E2, E3};                                                                          \
        double v_[4];                                                                                                     \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) v_[u_] = (ABL & 4) ? (double)e_[u_] : tab[e_[u_] >> kPsRowBits];    \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                                 \
            if (ABL & 1) asm volatile("" ::"v"(v_[u_]), "v"(e_[u_]));                                                      \
            else __hip_atomic_fetch_add(&sums[e_[u_] & ((1u << kPsRowBits) - 1)], v_[u_], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); \
        }                                                                                                                 \
    } while (0)

// ABL (CANOPY_ABLATIONS builds only; wrong results): 1 no LDS adds, 2 no table swap, 4 no table reads, 8 no epilogue
template <bool MULTI, int ABL = 0> // MULTI: more than one column part (the table part in LDS changes); else the whole table stays resident
__global__ __launch_bounds__(kPsWaves * kWave) void k_ps(const uint32_t* __restrict__ stream, const uint32_t* __restrict__ goff,
                                                         const double* __restrict__ logp, const int64_t* __restrict__ row_ptr,
                                                         PsScalars sc, double* __restrict__ block_partials,
                                                         double* __restrict__ row_logp_out, uint64_t* __restrict__ keep_mask_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    double* tab = reinterpret_cast<double*>(smem);
    double* sums = reinterpret_cast<double*>(smem + kPsTabBytes) + (size_t)wave * kPsSlots;
    for (int i = lane; i < kPsSlots; i += kWave) sums[i] = 0.0;
    if (tid < 2) tab[kPsPart + tid] = 0.0; // the zero slot padding entries read

    Acc acc;
    acc_init(acc);
    const int H = sc.H;
    const int64_t G = (int64_t)gridDim.x;
    const int64_t n_ss = (sc.n_ws + kPsWaves - 1) / kPsWaves;
    const uint32_t pad = ps_pad_entry(lane);

    // the H + 1 group offsets of a wave-slab in ONE register: lane h holds goff[ws * H + h]
    auto load_offsets = [&](int64_t ss_) -> uint32_t {
        const int64_t ws_ = ss_ * kPsWaves + wave;
        return (ss_ < n_ss && ws_ < sc.n_ws && lane <= H) ? goff[ws_ * H + lane] : 0u;
    };
    // the table part of 8192 columns: 16 bytes per thread, 8 times (zeros beyond the last column)
    double2 tq[8], tr[8]; // the next part of the table, and the one after it (two parts of look-ahead, like the stream)
    // (every logp table carries two doubles of slack behind its last column: the 16-byte load of an odd last column is valid)
    auto load_table = [&](int h_) { // no branch around a load: clamped addresses, the values selected afterwards
        const int64_t c0 = (int64_t)h_ * kPsPart;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t c = c0 + k * 1024 + tid * 2;
            const bool in1 = c < sc.n_events, in2 = c + 1 < sc.n_events;
            double2 v = *reinterpret_cast<const double2*>(logp + (in1 ? c : 0));
            v.x = in1 ? v.x : 0.0;
            v.y = in2 ? v.y : 0.0;
            tr[k] = v;
        }
    };
    auto store_table = [&]() { // the older set goes into LDS, the younger one moves up
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<double2*>(tab + k * 1024 + tid * 2) = tq[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) tq[k] = tr[k];
    };

    // Register sets of stream groups: the part being added, and the two parts after it (their loads are in flight).
    uint32_t qa[kPsGP], qb[kPsGP], qc[kPsGP];
    int na = 0, nb = 0, nc = 0;       // groups of the run each set belongs to (wave-uniform)
    uint32_t ga = 0, gb = 0, gc = 0;  // first group of that run
    // position two parts ahead of the current one: slab-set f_ss (0, 1 or 2 slab-sets ahead), part f_h
    int64_t ss = blockIdx.x;
    uint32_t off0 = load_offsets(ss), off1 = load_offsets(ss + G), off2 = load_offsets(ss + 2 * G);
    uint32_t off3 = load_offsets(ss + 3 * G); // in flight for a whole slab-set: nothing reads it before it has moved down
    int64_t f_ss = ss;
    int f_h = 0;
    int f_rel = 0; // (f_ss - ss) / G
#define PS_ISSUE(QF, NF, GF)                                                                                              \
    do { /* loads of the run at the look-ahead position into set F; then the position moves on by one part */           \
        const uint32_t offv_ = f_rel == 0 ? off0 : (f_rel == 1 ? off1 : off2);                                            \
        const uint32_t g0_ = ps_lane_u32(offv_, f_h), g1_ = ps_lane_u32(offv_, f_h + 1);                                   \
        GF = g0_;                                                                                                         \
        NF = f_ss < n_ss ? (int)(g1_ - g0_) : 0;                                                                          \
        /* unconditional loads (a branch around a load makes the compiler drain the memory queue): positions past the  \
           run's end re-read its last group (a cache hit), never used */                                               \
        _Pragma("unroll") for (int k_ = 0; k_ < kPsGP; ++k_)                                                              \
            QF[k_] = stream[((size_t)g0_ + (k_ < NF ? k_ : (NF > 0 ? NF - 1 : 0))) * kGroup + lane];                       \
        if (++f_h == H) { f_h = 0; f_ss += G; ++f_rel; }                                                                   \
    } while (0)
#define PS_ADD(QC, NC, GC)                                                                                                \
    do { /* every group of the current run: the first kPsGP out of registers, a longer run's rest straight from memory */ \
        _Pragma("unroll") for (int k_ = 0; k_ < kPsGP; k_ += 4)                                                           \
            if (k_ < NC)                                                                                                  \
                PS_GROUP4(QC[k_], k_ + 1 < NC ? QC[k_ + 1] : pad, k_ + 2 < NC ? QC[k_ + 2] : pad, k_ + 3 < NC ? QC[k_ + 3] : pad); \
        for (int k_ = kPsGP; k_ < NC; k_ += 4) {                                                                          \
            uint32_t x_[4];                                                                                               \
            _Pragma("unroll") for (int u2_ = 0; u2_ < 4; ++u2_)                                                           \
                x_[u2_] = k_ + u2_ < NC ? stream[((size_t)GC + k_ + u2_) * kGroup + lane] : pad;                           \
            PS_GROUP4(x_[0], x_[1], x_[2], x_[3]);                                                                        \
        }                                                                                                                 \
    } while (0)
#define PS_PART(QC, NC, GC, QF, NF, GF)                                                                                   \
    do {                                                                                                                  \
        if (swap_table && !(ABL & 2)) {                                                                                   \
            __syncthreads(); /* every wave has finished reading the previous part of the table */                         \
            store_table();                                                                                                \
            __syncthreads();                                                                                              \
        }                                                                                                                 \
        if (MULTI) { /* the part of the table two parts ahead into registers (a few times too often at the very end) */     \
            load_table((h + 2) % H);                                                                                      \
            swap_table = h + 1 < H || ss + G < n_ss;                                                                      \
        } else swap_table = false;                                                                                        \
        PS_ISSUE(QF, NF, GF);                                                                                             \
        PS_ADD(QC, NC, GC);                                                                                               \
        if (++h == H) { /* the slab-set is complete: fold this wave's 1024 rows, move on */                               \
            const int64_t ws_ = ss * kPsWaves + wave;                                                                     \
            if (ws_ < sc.n_ws && !(ABL & 8))                                                                              \
                ws_epilogue<kPsRows>(sums, ws_, lane, &acc, sc.n_rows, sc.row_offset, sc.log_cut, sc.limit_order, sc.need_len, \
                                     row_ptr, row_logp_out, keep_mask_out, 0, kPsRows / kWave);                           \
            h = 0;                                                                                                        \
            ss += G;                                                                                                      \
            off0 = off1; off1 = off2; off2 = off3; off3 = load_offsets(ss + 3 * G);                                       \
            --f_rel;                                                                                                      \
        }                                                                                                                 \
    } while (0)

    if (ss < n_ss) {
        bool swap_table = true; // the first part always goes into LDS
        load_table(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) tq[k] = tr[k];
        load_table(H > 1 ? 1 : 0);
        int h = 0;
        PS_ISSUE(qa, na, ga); // the first two runs; from here on every part issues the run two parts ahead
        PS_ISSUE(qb, nb, gb);
        while (ss < n_ss) {
            PS_PART(qa, na, ga, qc, nc, gc);
            if (ss >= n_ss) break;
            PS_PART(qb, nb, gb, qa, na, ga);
            if (ss >= n_ss) break;
            PS_PART(qc, nc, gc, qb, nb, gb);
        }
    }
#undef PS_PART
#undef PS_ADD
#undef PS_ISSUE

    // workgroup reduce of the wave records (the sums slices are dead now)
    __syncthreads();
    double* s_red = reinterpret_cast<double*>(smem);
    acc_wave_reduce(acc);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    __syncthreads();
    if (tid == 0) {
        Acc r;
        acc_load(r, s_red);
        for (int w = 1; w < kPsWaves; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(r, o);
        }
        acc_store(block_partials + (size_t)blockIdx.x * kAccDoubles, r);
    }
}
#undef PS_GROUP4

} // namespace

int ps_parts(int32_t n_events) { return n_events <= 0 ? 1 : (int)(((int64_t)n_events + kPsPart - 1) / kPsPart); }
int ps_max_events() { return kPsMaxParts * kPsPart; }
int ps_default_grid(int n_cu, int64_t n_ws) {
    int64_t g = (n_ws + kPsWaves - 1) / kPsWaves;
    if (g > n_cu) g = n_cu; // one workgroup per CU: its LDS holds a table part and 8192 row sums
    return g < 1 ? 1 : (int)g;
}

// Builds the part stream of the model's canonical CSR. *ok = false: the model does not fit (too many parts, absurd rows,
// more than 2^32 groups): the caller keeps another variant.
hipError_t ps_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int32_t n_events, int32_t max_row_len,
                    uint32_t** d_stream_out, uint32_t** d_goff_out, int64_t* total_groups_out, int64_t* n_ws_out, bool* ok,
                    hipStream_t s) {
    *ok = false;
    *d_stream_out = nullptr; *d_goff_out = nullptr; *total_groups_out = 0;
    const int H = ps_parts(n_events);
    const int64_t n_ws = (n_rows + kPsRows - 1) / kPsRows;
    *n_ws_out = n_ws;
    if (H > kPsMaxParts) return hipSuccess;
    const int64_t n_tasks = n_ws * H;
    int64_t gcap64 = ((int64_t)kPsRows * (max_row_len > 0 ? max_row_len : 1) + kGroup - 1) / kGroup + (max_row_len > 0 ? max_row_len : 1) + 2;
    if (gcap64 > 4096 || (double)gcap64 * (double)(n_tasks > 0 ? n_tasks : 1) > 3.0e9) return hipSuccess; // absurd rows / scratch
    const int gcap = (int)gcap64;
    uint8_t* d_cnt = nullptr;
    int64_t* d_off = nullptr;
    uint32_t* d_goff = nullptr;
    int* d_ovf = nullptr;
    uint32_t* d_stream = nullptr;
    auto bail = [&](hipError_t err) {
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_off) (void)hipFree(d_off);
        if (d_goff) (void)hipFree(d_goff);
        if (d_ovf) (void)hipFree(d_ovf);
        if (d_stream) (void)hipFree(d_stream);
        return err;
    };
    hipError_t e = hipMalloc(&d_cnt, (size_t)(n_tasks > 0 ? n_tasks : 1) * gcap);
    if (e == hipSuccess) e = hipMalloc(&d_off, sizeof(int64_t) * (size_t)(n_tasks + 1));
    if (e == hipSuccess) e = hipMalloc(&d_goff, sizeof(uint32_t) * (size_t)(n_tasks + 1 + kWave)); // (+64: a wave reads lanes 0..H of its slice)
    if (e == hipSuccess) e = hipMalloc(&d_ovf, sizeof(int));
    if (e == hipSuccess) e = hipMemsetAsync(d_ovf, 0, sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_off, 0, sizeof(int64_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_goff, 0, sizeof(uint32_t) * (size_t)(n_tasks + 1 + kWave), s);
    if (e != hipSuccess) return bail(e);
    int64_t total = 0;
    int ovf = 0;
    if (n_tasks > 0) {
        hipLaunchKernelGGL((k_ps_build<false>), dim3(grid_for(n_tasks, 64)), dim3(64), 0, s, d_row_ptr, d_col_idx, n_rows, H, n_tasks, gcap,
                           d_cnt, d_off + 1, (const int64_t*)nullptr, (uint32_t*)nullptr, d_ovf);
        e = hipGetLastError();
        if (e == hipSuccess) e = scan_inclusive_i64(d_off + 1, n_tasks, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&ovf, d_ovf, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_off + n_tasks, sizeof(int64_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return bail(e);
        if (ovf || total >= ((int64_t)1 << 32) - kPsGP) { (void)bail(hipSuccess); return hipSuccess; }
    }
    const int64_t n_entries = (total + kPsGP) * kGroup; // (+ padding groups a predicated look-ahead may touch)
    e = hipMalloc(&d_stream, sizeof(uint32_t) * (size_t)n_entries);
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL(k_ps_fill, dim3(grid_for(n_entries, 256) > 4096 ? 4096 : grid_for(n_entries, 256)), dim3(256), 0, s, d_stream, n_entries);
    if (n_tasks > 0) {
        hipLaunchKernelGGL((k_ps_build<true>), dim3(grid_for(n_tasks, 64)), dim3(64), 0, s, d_row_ptr, d_col_idx, n_rows, H, n_tasks, gcap,
                           d_cnt, (int64_t*)nullptr, (const int64_t*)d_off, d_stream, d_ovf);
        hipLaunchKernelGGL(k_ps_narrow, dim3(grid_for(n_tasks + 1, 256) > 4096 ? 4096 : grid_for(n_tasks + 1, 256)), dim3(256), 0, s,
                           (const int64_t*)d_off, d_goff, n_tasks + 1);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    (void)hipFree(d_cnt);
    (void)hipFree(d_off);
    (void)hipFree(d_ovf);
    *d_stream_out = d_stream;
    *d_goff_out = d_goff;
    *total_groups_out = total;
    *ok = true;
    return hipSuccess;
}

hipError_t launch_ps_kernel(const RowKernelArgs& ra, const uint32_t* stream, const uint32_t* goff, int64_t n_ws, int need_len,
                            int grid, hipStream_t s) {
    PsScalars sc;
    sc.n_rows = ra.n_rows; sc.n_ws = n_ws; sc.row_offset = ra.row_offset;
    sc.log_cut = ra.log_cut; sc.limit_order = ra.limit_order; sc.need_len = need_len;
    sc.H = ps_parts(ra.n_events); sc.n_events = ra.n_events;
#define PS_GO(M, A)                                                                                                       \
    do { /* (the LDS attribute is per device: set on every launch) */                                                     \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_ps<M, A>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (e_ != hipSuccess) return e_;                                                                                  \
        hipLaunchKernelGGL((k_ps<M, A>), dim3(grid), dim3(kPsWaves * kWave), kPsLdsBytes, s, stream, goff, ra.logp, ra.row_ptr, sc,  \
                           ra.block_partials, ra.row_logp_out, ra.keep_mask_out);                                         \
        return hipGetLastError();                                                                                         \
    } while (0)
#ifdef CANOPY_ABLATIONS // timing ablations of benchmarks/ (wrong results)
    static const int dbg = CANOPY_TUNE_ENV("CANOPY_PS_DEBUG") ? atoi(CANOPY_TUNE_ENV("CANOPY_PS_DEBUG")) : 0;
    if (sc.H > 1) {
        if (dbg == 1) PS_GO(true, 1);
        if (dbg == 2) PS_GO(true, 2);
        if (dbg == 4) PS_GO(true, 4);
        if (dbg == 5) PS_GO(true, 5);
        if (dbg == 7) PS_GO(true, 7);
        if (dbg == 8) PS_GO(true, 8);
        if (dbg == 15) PS_GO(true, 15);
    }
#endif
    if (sc.H > 1) PS_GO(true, 0);
    PS_GO(false, 0);
#undef PS_GO
}

} // namespace canopy

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

// (the slab epilogue -- ws_epilogue -- lives in canopy_device.h: the band-stream kernel folds its slabs with it too)

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

} // namespace canopy

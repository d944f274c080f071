// kernels_vs.hip -- the "value-scatter" (VS) variant of K1..K4 for large basic-event counts,
// gfx950 / wave64 only.
//
// The wave-slab stream kernel (kernels_ws.hip) keeps row sums in LDS and is limited to one worker wave
// per SIMD (16 K row sums fill the LDS), which leaves every latency exposed (profiles/r01_pmc_summary.txt).
// VS keeps the *gathered values* in LDS instead and needs no ordering between waves at all:
//
//   * rows are cut into slabs of consecutive rows with at most kSlabCap entries (one slab = one
//     workgroup iteration, 16 waves);
//   * phase 1: the slab's entries arrive as a column-ordered stream of packed u32 (col << 14 | dest),
//     dest = the entry's CSR position inside the slab. Every wave takes chunks of the stream, gathers
//     logp[col] -- 64 neighbouring columns per instruction, cache lines shared by all 16 waves -- and
//     stores the value with ds_write_b64 into vals[dest]. Any order, any wave: each entry has its own slot.
//   * phase 2: lane per row. The lane walks its row's slots vals[b..e) left to right and adds with plain
//     v_add_f64: exactly the CSR-order chain of the oracle -> s_i bit-identical. Then the shared row
//     epilogue; kept rows go through a wave-local LDS queue so that exp/log1p run 64 rows at a time.
//
// HBM traffic per cut set: 4 B per entry (stream) + 8 B (row_ptr) = the canonical 32 B at avg 6.
#include <stdlib.h>

#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

constexpr int kDestBits = 14;
constexpr uint32_t kDestMask = (1u << kDestBits) - 1;
constexpr int kSlabCap = (1 << kDestBits) - 1; // real entries per slab; slot kSlabCap is the no-op sink
constexpr int kVsChunk = 256;
constexpr int kVsWaves = 16;
constexpr int kVsThreads = kVsWaves * kWave;
constexpr int kVsMaxChunksPerWave = 5;  // ceil((kSlabCap/64 + slack groups)/4/16)
constexpr int kVsRowsPre = 4;           // rows per thread whose row_ptr pair is preloaded
constexpr int kQueue = 128;             // per-wave kept-row queue (doubles)

struct VsLds {
    static constexpr size_t vals_bytes = (size_t)(kSlabCap + 1) * 8;
    static constexpr size_t queue_bytes = (size_t)kVsWaves * kQueue * 8;
    static constexpr size_t red_bytes = (size_t)kVsWaves * kAccDoubles * 8;
    static constexpr size_t total = vals_bytes + queue_bytes + red_bytes;
};
static_assert(VsLds::total <= 160 * 1024, "LDS budget");

// ---- builder ---------------------------------------------------------------------------------------

// slab s starts at the first row r with row_ptr[r] >= s * target (target = kSlabCap - max_row_len),
// so every slab holds at most kSlabCap entries; independent binary searches, one per slab.
__global__ __launch_bounds__(256) void k_vs_slab_rows(const int64_t* __restrict__ row_ptr, int64_t n_rows,
                                                      int64_t target, int64_t n_slabs, int64_t* slab_row) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s > n_slabs) return;
    if (s == n_slabs) { slab_row[s] = n_rows; return; }
    const int64_t want = s * target;
    int64_t lo = 0, hi = n_rows;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (row_ptr[mid] >= want) hi = mid; else lo = mid + 1;
    }
    slab_row[s] = lo;
}

// One thread per slab: greedy placement of the slab's entries into 64-entry groups by column
// (target group = the column's share of the slab's group count, spill forward when full).
template <bool WRITE>
__global__ __launch_bounds__(64) void k_vs_build(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                 int32_t n_events, const int64_t* __restrict__ slab_row, int64_t n_slabs,
                                                 int gcap, uint8_t* cnt_scratch, int64_t* chunks,
                                                 const int64_t* __restrict__ vs_off, uint32_t* stream, int* overflow) {
    const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slabs) return;
    const int64_t r0 = slab_row[s], r1 = slab_row[s + 1];
    const int64_t k0 = row_ptr[r0], k1 = row_ptr[r1];
    const int64_t nnz = k1 - k0;
    int64_t ng_nom = (nnz + 63) / 64;
    if (ng_nom < 1) ng_nom = 1;
    uint8_t* cnt = cnt_scratch + (size_t)s * gcap;
    for (int g = 0; g < gcap; ++g) cnt[g] = 0;
    int g_max = -1;
    const int64_t base = WRITE ? vs_off[s] * kVsChunk : 0;
    bool ovf = nnz > kSlabCap;
    for (int64_t k = k0; k < k1 && !ovf; ++k) {
        const uint32_t c = (uint32_t)col_idx[k];
        int g = (int)(((uint64_t)c * (uint64_t)ng_nom) / (uint64_t)n_events);
        while (g < gcap && cnt[g] >= 64) ++g;
        if (g >= gcap) { ovf = true; break; }
        const int slot = cnt[g]++;
        if (WRITE) stream[base + (int64_t)(g >> 2) * kVsChunk + slot * 4 + (g & 3)] = (c << kDestBits) | (uint32_t)(k - k0);
        g_max = g > g_max ? g : g_max;
    }
    if (ovf) atomicExch(overflow, 1);
    if (!WRITE) chunks[s] = (int64_t)((g_max + 1 + 3) / 4);
}

__global__ __launch_bounds__(256) void k_vs_fill(uint32_t* p, int64_t n, uint32_t v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = v;
}

// ---- the kernel --------------------------------------------------------------------------------------

struct VsScalars {
    int64_t total_chunks; // the stream holds one no-op chunk after them
    int64_t n_rows, n_slabs, row_offset;
    double log_cut;
    int32_t limit_order;
};

__device__ __forceinline__ int64_t uload64(const int64_t* __restrict__ p) {
    const int64_t v = *p;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)v);
    const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)v >> 32));
    return (int64_t)(((uint64_t)hi << 32) | (uint64_t)lo);
}

__device__ __forceinline__ void vs_barrier() { // LDS drained, then the workgroup barrier (no vmcnt drain)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// exp / log1p / compensated sums over 64 queued kept rows at once
// (not inlined: keeps the transcendental code's registers out of the streaming loop)
struct FoldState { CSum rare, l1p; double n_certain; };
__device__ __noinline__ FoldState vs_fold(FoldState f, double s, bool on) {
    fold_kept_row(f.rare, f.l1p, f.n_certain, s, on);
    return f;
}
#define VS_FOLD(ACC, S, ON)                                            \
    do {                                                               \
        FoldState f_ = {ACC.rare, ACC.l1p, ACC.n_certain};             \
        f_ = vs_fold(f_, S, ON);                                       \
        ACC.rare = f_.rare; ACC.l1p = f_.l1p; ACC.n_certain = f_.n_certain; \
    } while (0)

template <int ABL>
__global__ __launch_bounds__(kVsThreads) void k_vs(const uint4* __restrict__ stream, const int64_t* __restrict__ vs_off,
                                                   const int64_t* __restrict__ slab_row,
                                                   const int64_t* __restrict__ row_ptr, const double* __restrict__ logp,
                                                   VsScalars sc, double* __restrict__ block_partials,
                                                   double* __restrict__ row_logp_out,
                                                   uint64_t* __restrict__ keep_mask_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* vals = reinterpret_cast<double*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    double* queue = reinterpret_cast<double*>(smem + VsLds::vals_bytes) + wave * kQueue;
    double* s_red = reinterpret_cast<double*>(smem + VsLds::vals_bytes + VsLds::queue_bytes);
    int n_queued = 0; // wave-uniform

    Acc acc;
    acc_init(acc);

    const int64_t G = (int64_t)gridDim.x;
    int64_t s = (int64_t)blockIdx.x; // workgroup-uniform
    uint4 q[kVsMaxChunksPerWave];
    // this wave's chunks of slab s: c0 + wave, c0 + wave + 16, ...; chunks past the slab's end read the
    // trailing no-op chunk (col 0 -> sink slot)
#define VS_LOAD_CHUNKS(SLAB)                                                                              \
    do {                                                                                                  \
        int64_t c0_ = sc.total_chunks, c1_ = sc.total_chunks;                                             \
        if ((SLAB) < sc.n_slabs) { c0_ = uload64(vs_off + (SLAB)); c1_ = uload64(vs_off + (SLAB) + 1); }  \
        _Pragma("unroll") for (int k = 0; k < kVsMaxChunksPerWave; ++k) {                                 \
            const int64_t c = c0_ + wave + (int64_t)k * kVsWaves;                                         \
            q[k] = stream[(c < c1_ ? c : sc.total_chunks) * kWave + lane];                                \
        }                                                                                                 \
    } while (0)
    VS_LOAD_CHUNKS(s);

    for (; s < sc.n_slabs; s += G) {
        // ---------------- phase 1: gather in column order, scatter values to CSR slots ----------------
        // rows of this slab handled by this thread in phase 2: R0 + tid + 1024*i
        const int64_t R0 = uload64(slab_row + s), R1 = uload64(slab_row + s + 1);
        const int64_t kbase = uload64(row_ptr + R0);
        int rb[kVsRowsPre], rl[kVsRowsPre]; // slot of the row's first entry, row length
        {
            // two chunks (8 gathers per lane) at a time keeps the register budget of a 16-wave workgroup
            double v[2][4];
#pragma unroll
            for (int k0 = 0; k0 < kVsMaxChunksPerWave; k0 += 2) {
#pragma unroll
                for (int k = k0; k < k0 + 2 && k < kVsMaxChunksPerWave; ++k) {
                    const uint32_t e[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[k - k0][u] = (ABL & 1) ? (double)e[u] : logp[e[u] >> kDestBits];
                }
                if (k0 == 0) { // the row_ptr pairs are requested behind the first gathers
#pragma unroll
                    for (int i = 0; i < kVsRowsPre; ++i) {
                        const int64_t r = R0 + tid + (int64_t)i * kVsThreads;
                        int64_t b = kbase, e = kbase;
                        if (r < R1) { b = row_ptr[r]; e = row_ptr[r + 1]; }
                        rb[i] = (int)(b - kbase);
                        rl[i] = (int)(e - b);
                    }
                }
#pragma unroll
                for (int k = k0; k < k0 + 2 && k < kVsMaxChunksPerWave; ++k) {
                    const uint32_t e[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) vals[e[u] & kDestMask] = v[k - k0][u];
                }
            }
        }
        vs_barrier();
        VS_LOAD_CHUNKS(s + G); // next slab's stream chunks fly under phase 2

        // ---------------- phase 2: lane per row, sequential CSR-order sums out of LDS ----------------
        for (int64_t r0 = R0; r0 < R1 && !(ABL & 2); r0 += (int64_t)kVsRowsPre * kVsThreads) {
#pragma unroll
            for (int i = 0; i < kVsRowsPre; ++i) {
                const int64_t r = r0 + tid + (int64_t)i * kVsThreads;
                if (r0 + (int64_t)i * kVsThreads >= R1) break; // workgroup-uniform
                const bool valid = r < R1;
                int boff, len;
                if (r0 == R0) { boff = rb[i]; len = rl[i]; } // preloaded
                else { // slabs of very short rows: more than kVsRowsPre rows per thread
                    int64_t b = kbase, e = kbase;
                    if (valid) { b = row_ptr[r]; e = row_ptr[r + 1]; }
                    boff = (int)(b - kbase);
                    len = (int)(e - b);
                }
                const double* c = vals + boff;
                double sum = 0.0;
                for (int j0 = 0; j0 < len; j0 += 4) {
                    double w[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (j0 + u < len) w[u] = c[j0 + u];
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (j0 + u < len) sum = sum + w[u];
                }
                const bool keep = valid && (sum >= sc.log_cut) && (sc.limit_order == 0 || len <= sc.limit_order);
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    acc.n_kept += 1.0;
                    if (sum > acc.max_logp) { acc.max_logp = sum; acc.argmax = (double)(sc.row_offset + r); }
                    const int pos = n_queued + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                              __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    queue[pos] = sum;
                }
                n_queued += __builtin_popcountll(m);
                if (row_logp_out != nullptr && valid) row_logp_out[r] = sum;
                if (keep_mask_out != nullptr && lane == 0 && r0 + (int64_t)i * kVsThreads + wave * kWave < R1) {
                    // rows of a slab start at an arbitrary row: the 64 rows of this wave-iteration may straddle
                    // two mask words; both are owned by neighbouring iterations too -> combine with atomics
                    const int64_t rw = r0 + (int64_t)i * kVsThreads + wave * kWave; // row of lane 0
                    const int sh = (int)(rw & 63);
                    atomicOr(reinterpret_cast<unsigned long long*>(keep_mask_out) + (rw >> 6), m << sh);
                    if (sh != 0 && (m >> (64 - sh)) != 0)
                        atomicOr(reinterpret_cast<unsigned long long*>(keep_mask_out) + (rw >> 6) + 1, m >> (64 - sh));
                }
                if (n_queued >= kWave) { // 64 kept rows at once through exp / log1p
                    VS_FOLD(acc, queue[lane], true);
                    if (lane < n_queued - kWave) queue[lane] = queue[kWave + lane];
                    n_queued -= kWave;
                }
            }
        }
        vs_barrier(); // vals is rewritten by the next slab
    }
#undef VS_LOAD_CHUNKS
    if (n_queued > 0) VS_FOLD(acc, (lane < n_queued ? queue[lane] : 0.0), lane < n_queued);

    acc_wave_reduce(acc);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    vs_barrier();
    if (tid == 0) {
        Acc r;
        acc_load(r, s_red);
        for (int w = 1; w < kVsWaves; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(r, o);
        }
        acc_store(block_partials + (size_t)blockIdx.x * kAccDoubles, r);
    }
}

int grid_for(int64_t n, int block) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)g;
}

} // namespace

int vs_max_events() { return 1 << (32 - kDestBits); }
int vs_max_row_len() { return kSlabCap / 4; }

int vs_default_grid(int n_cu, int64_t n_slabs) {
    int64_t g = n_slabs < n_cu ? n_slabs : n_cu;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t vs_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int64_t nnz, int32_t n_events,
                    int32_t max_row_len, uint32_t** d_stream_out, int64_t** d_vs_off_out, int64_t** d_slab_row_out,
                    int64_t* total_chunks_out, int64_t* n_slabs_out, bool* ok, hipStream_t s) {
    *ok = false;
    if (max_row_len > vs_max_row_len() || n_events > vs_max_events()) return hipSuccess;
    const int64_t target = kSlabCap - (max_row_len > 0 ? max_row_len : 1);
    const int64_t n_slabs = n_rows > 0 ? (nnz + target - 1) / target + 1 : 0; // + 1: trailing rows without entries
    *n_slabs_out = n_slabs;
    int64_t* d_slab_row = nullptr;
    int64_t* d_off = nullptr;
    uint8_t* d_cnt = nullptr;
    int* d_ovf = nullptr;
    uint32_t* d_stream = nullptr;
    auto bail = [&](hipError_t err) {
        if (d_slab_row) (void)hipFree(d_slab_row);
        if (d_off) (void)hipFree(d_off);
        if (d_cnt) (void)hipFree(d_cnt);
        if (d_ovf) (void)hipFree(d_ovf);
        if (d_stream) (void)hipFree(d_stream);
        return err;
    };
    const int gcap = kSlabCap / 64 + 1 + 48; // nominal groups + spill room
    hipError_t e = hipMalloc(&d_slab_row, sizeof(int64_t) * (size_t)(n_slabs + 1));
    if (e == hipSuccess) e = hipMalloc(&d_off, sizeof(int64_t) * (size_t)(n_slabs + 1));
    if (e == hipSuccess) e = hipMalloc(&d_cnt, (size_t)(n_slabs > 0 ? n_slabs : 1) * gcap);
    if (e == hipSuccess) e = hipMalloc(&d_ovf, sizeof(int));
    if (e != hipSuccess) return bail(e);
    e = hipMemsetAsync(d_ovf, 0, sizeof(int), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_off, 0, sizeof(int64_t), s);
    if (e != hipSuccess) return bail(e);
    hipLaunchKernelGGL(k_vs_slab_rows, dim3(grid_for(n_slabs + 1, 256)), dim3(256), 0, s, d_row_ptr, n_rows, target, n_slabs,
                       d_slab_row);
    int ovf = 0;
    int64_t total = 0;
    if (n_slabs > 0) {
        hipLaunchKernelGGL((k_vs_build<false>), dim3(grid_for(n_slabs, 64)), dim3(64), 0, s, d_row_ptr, d_col_idx, n_events,
                           (const int64_t*)d_slab_row, n_slabs, gcap, d_cnt, d_off + 1, (const int64_t*)nullptr,
                           (uint32_t*)nullptr, d_ovf);
        e = hipGetLastError();
        if (e == hipSuccess) e = scan_inclusive_i64(d_off + 1, n_slabs, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&ovf, d_ovf, sizeof(int), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&total, d_off + n_slabs, sizeof(int64_t), hipMemcpyDeviceToHost, s);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    if (ovf) { (void)bail(hipSuccess); return hipSuccess; }
    const int64_t n_entries = (total + 1) * kVsChunk; // + the trailing no-op chunk
    e = hipMalloc(&d_stream, sizeof(uint32_t) * (size_t)n_entries);
    if (e != hipSuccess) return bail(e);
    int fg = grid_for(n_entries, 256);
    if (fg > 4096) fg = 4096;
    hipLaunchKernelGGL(k_vs_fill, dim3(fg), dim3(256), 0, s, d_stream, n_entries, (uint32_t)kSlabCap); // col 0 -> sink
    if (n_slabs > 0) {
        hipLaunchKernelGGL((k_vs_build<true>), dim3(grid_for(n_slabs, 64)), dim3(64), 0, s, d_row_ptr, d_col_idx, n_events,
                           (const int64_t*)d_slab_row, n_slabs, gcap, d_cnt, (int64_t*)nullptr, (const int64_t*)d_off,
                           d_stream, d_ovf);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return bail(e);
    (void)hipFree(d_cnt);
    (void)hipFree(d_ovf);
    *d_stream_out = d_stream;
    *d_vs_off_out = d_off;
    *d_slab_row_out = d_slab_row;
    *total_chunks_out = total;
    *ok = true;
    return hipSuccess;
}

hipError_t launch_vs_kernel(const RowKernelArgs& ra, const uint32_t* stream, const int64_t* vs_off,
                            const int64_t* slab_row, int64_t total_chunks, int64_t n_slabs, int grid, hipStream_t s) {
    VsScalars sc;
    sc.total_chunks = total_chunks;
    sc.n_rows = ra.n_rows; sc.n_slabs = n_slabs; sc.row_offset = ra.row_offset;
    sc.log_cut = ra.log_cut; sc.limit_order = ra.limit_order;
#define VS_LAUNCH(A)                                                                                                   \
    do { /* (the LDS attribute is per device and cheap to set: every launch) */                                        \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vs<A>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        if (e_ != hipSuccess) return e_;                                                                               \
        hipLaunchKernelGGL(k_vs<A>, dim3(grid), dim3(kVsThreads), VsLds::total, s, reinterpret_cast<const uint4*>(stream), vs_off, \
                           slab_row, ra.row_ptr, ra.logp, sc, ra.block_partials, ra.row_logp_out, ra.keep_mask_out);   \
        return hipGetLastError();                                                                                      \
    } while (0)
#ifdef CANOPY_ABLATIONS // timing ablations of benchmarks/ (wrong results)
    static const int dbg = CANOPY_TUNE_ENV("CANOPY_VS_DEBUG") ? atoi(CANOPY_TUNE_ENV("CANOPY_VS_DEBUG")) : 0;
    if (dbg == 1) VS_LAUNCH(1);
    if (dbg == 2) VS_LAUNCH(2);
    if (dbg == 3) VS_LAUNCH(3);
#endif
    VS_LAUNCH(0);
#undef VS_LAUNCH
}

} // namespace canopy

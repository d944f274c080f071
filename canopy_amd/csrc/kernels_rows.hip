// kernels_rows.hip -- K1..K4 fused: per-cut-set log-probability (segmented gather-sum over CSR
// rows), cut-off / limit-order filter, rare-event and MCUB partial sums, argmax, optional per-row
// outputs. gfx950 (MI355X) only, wave64.
//
// Fills the hole the reference leaves at src/canopy.cpp:1-31 (no quantification code); the
// approximations are the ones named at reference src/mef/openpsa/settings.h:19,22.
//
// Variant ROWS / ROWS_LDS ("lane per row", canonical CSR straight from HBM):
//   a WAVE owns 128 consecutive rows per iteration. Their col_idx span is one contiguous range,
//   streamed with 16-byte coalesced loads into the wave's private LDS staging area; each lane then
//   walks its own rows out of LDS left to right and adds logp[col] with plain v_add_f64 -- the fixed
//   CSR order is what makes s_i bit-identical to the CPU oracle. logp is gathered through L1/L2
//   (ROWS) or from an LDS-resident copy of the table (ROWS_LDS, E*8 B must fit). No workgroup
//   barriers in the loop: waves are independent, latency is hidden by 8-32 waves per CU.
#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

constexpr int kEntriesPerRowBudget = 12;  // LDS staging capacity per 64-row group = 64 * 12 col entries
constexpr int kTableLdsMaxEvents = 12288; // 96 KiB of logp at most in LDS
constexpr int kRowsWaves = 8;             // waves per workgroup
constexpr int kRowsK = 2;                 // 64-row groups per wave iteration
constexpr int kRowsQueue = 128;           // per-wave kept-row queue (doubles)

__host__ __device__ constexpr size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

struct RowsLds {
    static constexpr int cap = 64 * kRowsK * kEntriesPerRowBudget;          // entries per wave
    static constexpr size_t stage_bytes = align16((size_t)(cap + 16) * 4);  // per wave
    static constexpr size_t queue_bytes = (size_t)kRowsQueue * 8;           // per wave
    static constexpr size_t red_bytes = (size_t)kRowsWaves * kAccDoubles * 8;
    static size_t total(int table_events) {
        return kRowsWaves * (stage_bytes + queue_bytes) + red_bytes + align16((size_t)table_events * 8);
    }
};

struct RowsFold { CSum rare, l1p; double n_certain; };
// exp / log1p / compensated sums for 64 queued kept rows (not inlined: keeps the transcendental
// code's registers out of the streaming loop)
__device__ __noinline__ RowsFold rows_fold(RowsFold f, double s, bool on) {
    fold_kept_row(f.rare, f.l1p, f.n_certain, s, on);
    return f;
}

// Lane per row, WAVE-independent: a wave owns 64*kRowsK consecutive rows per iteration. Their
// col_idx span is contiguous; the wave streams it with 16-byte coalesced loads into its private LDS
// staging area (LDS executes one wave's operations in issue order, so no barrier is needed between
// the staging stores and the row walks), then every lane walks its rows left to right. Waves never
// wait for each other: all latency hiding comes from having 8-16 independent waves per CU.
template <bool TABLE_LDS>
__global__ __launch_bounds__(kRowsWaves * kWave) void k_rows(RowKernelArgs a) {
    using L = RowsLds;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    int32_t* stage = reinterpret_cast<int32_t*>(smem + (size_t)wave * L::stage_bytes);
    double* queue = reinterpret_cast<double*>(smem + kRowsWaves * L::stage_bytes + (size_t)wave * L::queue_bytes);
    double* s_red = reinterpret_cast<double*>(smem + kRowsWaves * (L::stage_bytes + L::queue_bytes));
    double* s_tab = s_red + kRowsWaves * kAccDoubles;

    if (TABLE_LDS) {
        for (int j = tid; j < a.n_events; j += kRowsWaves * kWave) s_tab[j] = a.logp[j];
        __syncthreads();
    }

    Acc acc;
    acc_init(acc);
    int n_queued = 0; // wave-uniform

    constexpr int64_t kIterRows = 64 * kRowsK;
    const int64_t GW = (int64_t)gridDim.x * kRowsWaves;
    const int64_t n_iters = (a.n_rows + kIterRows - 1) / kIterRows;
    // row_ptr of iteration it+GW is requested before iteration it is walked (one HBM latency instead
    // of two on the wave's critical path)
    int64_t b[kRowsK], e[kRowsK], lo = 0, hi = 0;
    auto load_rows = [&](int64_t it_) {
        const int64_t base_ = it_ * kIterRows;
#pragma unroll
        for (int k = 0; k < kRowsK; ++k) {
            const int64_t r = base_ + k * 64 + lane;
            b[k] = 0; e[k] = 0;
            if (r < a.n_rows) { b[k] = a.row_ptr[r]; e[k] = a.row_ptr[r + 1]; }
        }
        const int64_t r_end_ = base_ + kIterRows < a.n_rows ? base_ + kIterRows : a.n_rows;
        // same address in every lane; made scalar only where it is used, so the load stays in flight
        if (base_ < a.n_rows) { lo = a.row_ptr[base_]; hi = a.row_ptr[r_end_]; }
    };
    const int64_t it0 = (int64_t)blockIdx.x * kRowsWaves + wave;
    if (it0 < n_iters) load_rows(it0);
    for (int64_t it = it0; it < n_iters; it += GW) {
        const int64_t base = it * kIterRows;
        int64_t cb[kRowsK], ce[kRowsK];
#pragma unroll
        for (int k = 0; k < kRowsK; ++k) { cb[k] = b[k]; ce[k] = e[k]; }
        const int64_t clo = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)lo >> 32)) << 32) |
                                      (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)lo));
        const int64_t chi = (int64_t)(((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)hi >> 32)) << 32) |
                                      (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)hi));
        const int64_t lo4 = clo & ~int64_t(3); // 16-byte aligned start of the span
        const bool staged = (chi - lo4) <= (int64_t)L::cap; // wave-uniform
        if (staged) {
            for (int64_t k = lo4 + (int64_t)lane * 4; k < chi; k += 256) {
                const int4 v = *reinterpret_cast<const int4*>(a.col_idx + k);
                *reinterpret_cast<int4*>(stage + (k - lo4)) = v;
            }
        }
        if (it + GW < n_iters) load_rows(it + GW);
#pragma unroll
        for (int k = 0; k < kRowsK; ++k) {
            const int64_t r = base + k * 64 + lane;
            const bool valid = r < a.n_rows;
            const int len = (int)(ce[k] - cb[k]);
            double s = 0.0;
            if (staged) {
                const int32_t* c = stage + (cb[k] - lo4);
                for (int j0 = 0; j0 < len; j0 += 8) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < len) {
                            const int idx = c[j0 + u];
                            v[u] = TABLE_LDS ? s_tab[idx] : a.logp[idx];
                        }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (j0 + u < len) s = s + v[u];
                }
            } else { // very long rows: walk global memory directly
                const int32_t* c = a.col_idx + cb[k];
                for (int j = 0; j < len; ++j) {
                    const int idx = c[j];
                    s = s + (TABLE_LDS ? s_tab[idx] : a.logp[idx]);
                }
            }
            const bool keep = valid && (s >= a.log_cut) && (a.limit_order == 0 || len <= a.limit_order);
            const unsigned long long m = __ballot(keep);
            if (m != 0) {
                if (keep) {
                    acc.n_kept += 1.0;
                    if (s > acc.max_logp) { acc.max_logp = s; acc.argmax = (double)(a.row_offset + r); }
                    queue[n_queued + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                                                    __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = s;
                }
                n_queued += __builtin_popcountll(m);
                if (n_queued >= kWave) { // 64 kept rows at once through exp / log1p
                    RowsFold f = {acc.rare, acc.l1p, acc.n_certain};
                    f = rows_fold(f, queue[lane], true);
                    acc.rare = f.rare; acc.l1p = f.l1p; acc.n_certain = f.n_certain;
                    if (lane < n_queued - kWave) queue[lane] = queue[kWave + lane];
                    n_queued -= kWave;
                }
            }
            if (a.row_logp_out != nullptr && valid) a.row_logp_out[r] = s;
            if (a.keep_mask_out != nullptr && lane == 0 && r < a.n_rows) a.keep_mask_out[r >> 6] = m;
        }
    }
    if (n_queued > 0) {
        RowsFold f = {acc.rare, acc.l1p, acc.n_certain};
        f = rows_fold(f, lane < n_queued ? queue[lane] : 0.0, lane < n_queued);
        acc.rare = f.rare; acc.l1p = f.l1p; acc.n_certain = f.n_certain;
    }

    acc_wave_reduce(acc);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    __syncthreads();
    if (tid == 0) {
        Acc t;
        acc_load(t, s_red);
        for (int w = 1; w < kRowsWaves; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(t, o);
        }
        acc_store(a.block_partials + (size_t)blockIdx.x * kAccDoubles, t);
    }
}

// Deterministic final reduce of the per-workgroup records (finalize_records in canopy_device.h).
// host_rec: null, or 16 doubles of pinned host memory -- a ticketed pass publishes its record there ([0..8), and the
// conservative pass's candidate total as a u64 at [8]), so that fetching it is an event wait and a memcpy, no device copy.
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ parts, int n, double* out,
                                                  double* gather, int world, int rank, unsigned long long* fr_count,
                                                  double* host_rec) {
    __shared__ double s_red[4 * kAccDoubles];
    __shared__ double s_out[kAccDoubles];
    finalize_records(parts, n, out, gather, world, rank, s_red, s_out);
    if (host_rec != nullptr && threadIdx.x < kAccDoubles) host_rec[threadIdx.x] = s_out[threadIdx.x];
    if (threadIdx.x == 0) {
        unsigned long long total = 0ULL;
        if (fr_count != nullptr) {       // FILTER_REFINE / FIXED_SUM: last pass's candidate total for the host,
            total = fr_count[0];         // accumulator cleared for the next pass's conservative kernel
            fr_count[1] = total;
            fr_count[0] = 0ULL;
        }
        if (host_rec != nullptr) reinterpret_cast<unsigned long long*>(host_rec)[kAccDoubles] = total;
    }
    if (host_rec != nullptr) __threadfence_system();
}

// the combined records behind the all-reduce of a ticketed pass, into pinned host memory
__global__ __launch_bounds__(64) void k_publish(const double* __restrict__ src, int n, double* host_dst) {
    for (int i = threadIdx.x; i < n; i += 64) host_dst[i] = src[i];
    __threadfence_system();
}

} // namespace

size_t rows_lds_table_limit_events() { return kTableLdsMaxEvents; }

int rows_kernel_default_grid(int n_cu, int variant, int64_t n_rows) {
    (void)variant;
    const int64_t n_iters = (n_rows + 64 * kRowsK - 1) / (64 * kRowsK);
    const int64_t need = (n_iters + kRowsWaves - 1) / kRowsWaves;
    int64_t g = (int64_t)n_cu * 4; // up to 4 workgroups (32 waves) resident per CU when the LDS allows
    if (g > need) g = need;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_rows_kernel(int variant, const RowKernelArgs& a, int grid, hipStream_t s) {
    // the attribute is per device and cheap to set: every launch (no process-wide "done" flag)
    hipError_t e = hipFuncSetAttribute(variant == CANOPY_KERNEL_ROWS_LDS ? reinterpret_cast<const void*>(&k_rows<true>)
                                                                         : reinterpret_cast<const void*>(&k_rows<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (variant == CANOPY_KERNEL_ROWS_LDS)
        hipLaunchKernelGGL((k_rows<true>), dim3(grid), dim3(kRowsWaves * kWave), RowsLds::total(a.n_events), s, a);
    else
        hipLaunchKernelGGL((k_rows<false>), dim3(grid), dim3(kRowsWaves * kWave), RowsLds::total(0), s, a);
    return hipGetLastError();
}

hipError_t launch_finalize(const double* block_partials, int n_blocks, double* out, double* gather,
                           int world, int rank, unsigned long long* fr_count, double* host_rec, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, s, block_partials, n_blocks, out, gather, world, rank, fr_count, host_rec);
    return hipGetLastError();
}

hipError_t launch_publish(const double* src, int n, double* host_dst, hipStream_t s) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, s, src, n, host_dst);
    return hipGetLastError();
}

} // namespace canopy

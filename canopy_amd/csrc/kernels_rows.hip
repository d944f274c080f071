// kernels_rows.hip -- K1..K4 fused: per-cut-set log-probability (segmented gather-sum over CSR
// rows), cut-off / limit-order filter, rare-event and MCUB partial sums, argmax, optional per-row
// outputs. gfx950 (MI355X) only, wave64.
//
// Fills the hole the reference leaves at src/canopy.cpp:1-31 (no quantification code); the
// approximations are the ones named at reference src/mef/openpsa/settings.h:19,22.
//
// Variant ROWS / ROWS_LDS ("lane per row"):
//   a workgroup owns a slab of BLOCK consecutive rows per iteration. Its col_idx span is one
//   contiguous range, streamed with 16-byte coalesced loads into LDS; each lane then walks its own
//   row out of LDS left to right and adds logp[col] with plain v_add_f64 -- the fixed CSR order is
//   what makes s_i bit-identical to the CPU oracle. logp is gathered through L2 (ROWS) or from an
//   LDS-resident copy of the table (ROWS_LDS, E*8 B must fit).
#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

namespace {

constexpr int kEntriesPerRowBudget = 12; // LDS staging capacity per slab = BLOCK * 12 col entries
constexpr int kTableLdsMaxEvents = 12288; // 96 KiB of logp at most in LDS

__host__ __device__ constexpr size_t align16(size_t x) { return (x + 15) & ~size_t(15); }

template <int BLOCK>
struct RowsLds {
    static constexpr int cap = BLOCK * kEntriesPerRowBudget;
    static constexpr size_t cols_bytes = align16((size_t)(cap + 16) * 4);
    static constexpr size_t span_bytes = 16;
    static constexpr size_t red_bytes = (size_t)(BLOCK / kWave) * kAccDoubles * 8;
    static size_t total(int table_events) {
        return cols_bytes + span_bytes + red_bytes + align16((size_t)table_events * 8);
    }
};

template <int BLOCK, bool TABLE_LDS>
__global__ __launch_bounds__(BLOCK) void k_rows(RowKernelArgs a) {
    using L = RowsLds<BLOCK>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* s_cols = reinterpret_cast<int32_t*>(smem);
    int64_t* s_span = reinterpret_cast<int64_t*>(smem + L::cols_bytes);
    double* s_red = reinterpret_cast<double*>(smem + L::cols_bytes + L::span_bytes);
    double* s_tab = reinterpret_cast<double*>(smem + L::cols_bytes + L::span_bytes + L::red_bytes);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid / kWave;

    if (TABLE_LDS) {
        for (int j = tid; j < a.n_events; j += BLOCK) s_tab[j] = a.logp[j];
        __syncthreads();
    }

    Acc acc;
    acc_init(acc);

    const int64_t n_slabs = (a.n_rows + BLOCK - 1) / BLOCK;
    for (int64_t slab = blockIdx.x; slab < n_slabs; slab += gridDim.x) {
        const int64_t r0 = slab * BLOCK;
        const int64_t r = r0 + tid;
        const bool valid = r < a.n_rows;
        int64_t b = 0, e = 0;
        if (valid) {
            b = a.row_ptr[r];
            e = a.row_ptr[r + 1];
        }
        const int64_t r_last = (r0 + BLOCK < a.n_rows ? r0 + BLOCK : a.n_rows) - 1;
        if (tid == 0) s_span[0] = b;
        if (r == r_last) s_span[1] = e;
        __syncthreads();
        const int64_t lo = s_span[0], hi = s_span[1];
        const int64_t lo4 = lo & ~int64_t(3); // 16-byte aligned start of the slab's col_idx span
        const bool staged = (hi - lo4) <= (int64_t)L::cap; // workgroup-uniform
        if (staged) {
            for (int64_t k = lo4 + (int64_t)tid * 4; k < hi; k += (int64_t)BLOCK * 4) {
                const int4 v = *reinterpret_cast<const int4*>(a.col_idx + k);
                *reinterpret_cast<int4*>(s_cols + (k - lo4)) = v;
            }
        }
        __syncthreads();

        const int len = (int)(e - b);
        double s = 0.0;
        if (staged) {
            const int32_t* c = s_cols + (b - lo4);
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
        } else { // a slab with very long rows: walk global memory directly
            const int32_t* c = a.col_idx + b;
            for (int j = 0; j < len; ++j) {
                const int idx = c[j];
                s = s + (TABLE_LDS ? s_tab[idx] : a.logp[idx]);
            }
        }

        const bool keep = row_epilogue(acc, s, len, valid, a.log_cut, a.limit_order,
                                       (double)(a.row_offset + r));
        if (a.row_logp_out != nullptr && valid) a.row_logp_out[r] = s;
        if (a.keep_mask_out != nullptr) {
            const unsigned long long m = __ballot(keep);
            if (lane == 0 && r < a.n_rows) a.keep_mask_out[r >> 6] = m;
        }
        __syncthreads(); // s_cols / s_span are rewritten by the next slab
    }

    acc_wave_reduce(acc);
    if (lane == 0) acc_store(s_red + wave * kAccDoubles, acc);
    __syncthreads();
    if (tid == 0) {
        Acc t;
        acc_load(t, s_red);
        for (int w = 1; w < BLOCK / kWave; ++w) {
            Acc o;
            acc_load(o, s_red + w * kAccDoubles);
            acc_merge(t, o);
        }
        acc_store(a.block_partials + (size_t)blockIdx.x * kAccDoubles, t);
    }
}

// Deterministic final reduce of the per-workgroup records: fixed thread <- record assignment,
// fixed tree. Output is the canopy_partial layout (hi/lo renormalised).
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ parts, int n, double* out,
                                                  double* gather, int world, int rank) {
    __shared__ double s_red[4 * kAccDoubles];
    __shared__ double s_out[kAccDoubles];
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
    // K5: own record into slot `rank`, zeros elsewhere -> a single sum all-reduce is an all-gather
    if (gather != nullptr)
        for (int i = threadIdx.x; i < world * kAccDoubles; i += 256)
            gather[i] = (i / kAccDoubles == rank) ? s_out[i % kAccDoubles] : 0.0;
}

} // namespace

size_t rows_lds_table_limit_events() { return kTableLdsMaxEvents; }

int rows_kernel_default_grid(int n_cu, int variant, int64_t n_rows) {
    const int block = 256;
    int per_cu = 8;
    (void)variant;
    const int64_t n_slabs = (n_rows + block - 1) / block;
    int64_t g = (int64_t)n_cu * per_cu;
    if (g > n_slabs) g = n_slabs;
    if (g < 1) g = 1;
    return (int)g;
}

hipError_t launch_rows_kernel(int variant, const RowKernelArgs& a, int grid, hipStream_t s) {
    constexpr int BLOCK = 256;
    if (variant == CANOPY_KERNEL_ROWS_LDS) {
        const size_t lds = RowsLds<BLOCK>::total(a.n_events);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rows<BLOCK, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            attr_set = true;
        }
        hipLaunchKernelGGL((k_rows<BLOCK, true>), dim3(grid), dim3(BLOCK), lds, s, a);
    } else {
        const size_t lds = RowsLds<BLOCK>::total(0);
        hipLaunchKernelGGL((k_rows<BLOCK, false>), dim3(grid), dim3(BLOCK), lds, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_finalize(const double* block_partials, int n_blocks, double* out, double* gather,
                           int world, int rank, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, s, block_partials, n_blocks, out, gather, world, rank);
    return hipGetLastError();
}

} // namespace canopy

// kernels_batch.hip -- batched quantification: K probability vectors ("trials") against ONE cut-set
// model, gfx950 / wave64 only, plus its C ABI entry points.
//
// Reference side: only the vocabulary exists -- uncertainty analysis draws num_trials probability vectors
// from the basic events' distributions (Settings::uncertainty_analysis / num_trials, reference
// src/mef/openpsa/settings.h:118-130,281-295; deviates src/mef/openpsa/expr/random_deviate.h) and
// quantifies the same cut sets under each. The caller supplies the K vectors; trial k's result is
// defined as exactly what canopy_quantify returns for p = p_trials[k] (same canonical arithmetic).
//
// Running K full passes would stream the model K times (0.49 ms each at config 3). Instead:
//   1. envelope  logp_max[j] = max_k logp[k][j]. IEEE addition is monotone in both operands, so the
//      CSR-order fp64 chain satisfies s_i(trial k) <= s_i(envelope) for every k, bit for bit: a row that
//      passes `s_i >= log(cut_off)` under ANY trial passes it under the envelope. No tolerance involved.
//   2. ONE ordinary quantification pass under the envelope (any kernel variant; FILTER_REFINE at scale)
//      with the keep mask requested = the union candidate set.
//   3. the candidate rows are compacted into a small CSR (row ids, offsets, columns);
//   4. k_batch_rows: lane per candidate row, T = 8 trials per lane. The log-probabilities are laid out
//      [event][trial], so one entry costs one 64-byte contiguous read for 8 trials; each trial's sum is
//      the same plain-add chain in CSR order as everywhere else, followed by the exact filter / exp /
//      log1p / count / argmax of that trial.
//   5. k_finalize_batch: deterministic reduce of the per-workgroup records, one workgroup per trial.
// With the default cut-off ~1 % of the rows are candidates: K trials cost one pass + K cheap passes over
// ~1 % of the model.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <new>
#include <thread>
#include <vector>

#include "canopy_device.h"
#include "canopy_internal.h"

using namespace canopy;

namespace canopy {

struct BatchState {
    double* d_in = nullptr;        // [chunk][E] staging of host logs
    int64_t in_cap = 0;
    double* d_logpT = nullptr;     // [E][kpad]
    int64_t logpT_cap = 0;
    double* d_logp_max = nullptr;  // [E]
    int64_t max_cap = 0;
    uint8_t* d_q8 = nullptr;       // [E rounded up to 16] + 16: the quantised filter's table of the envelope
    int64_t q8_cap = 0;
    uint64_t* d_fx = nullptr;      // [E] (a fixed-point envelope pass: cut_off 0)
    int64_t fx_cap = 0;
    int64_t* d_word_off = nullptr; // [n_words] inclusive scan of candidate counts per mask word
    int64_t word_cap = 0;
    int64_t* d_c_row = nullptr;    // [n_c] local row index of candidate i (ascending)
    int64_t c_row_cap = 0;
    int64_t* d_c_row_ptr = nullptr;// [n_c + 1]
    int64_t c_cap = 0;
    int32_t* d_c_col = nullptr;    // [nnz_c]
    int64_t col_cap = 0;
    int64_t* d_slab_kept = nullptr;// [n_slabs] kept candidates per slab -> exclusive prefix (fast path)
    int64_t* d_slab_nnz = nullptr; // [n_slabs] their entries -> exclusive prefix
    int64_t slab_cap = 0, slab_cap2 = 0;
    int64_t* d_totals = nullptr;   // [2] n_c, nnz_c
    int64_t* h_totals = nullptr;   // pinned
    double* d_parts = nullptr;     // [blocks][kpad][8]
    int64_t parts_cap = 0;
    double* d_out = nullptr;       // [kpad][8] canopy_partial records
    int64_t out_cap = 0;
    double* h_out = nullptr;       // pinned mirror
    int64_t h_out_cap = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double* h_stage[2] = {nullptr, nullptr}; // pinned: host logs of one chunk of trials; the copy of one runs while the other is filled
    int64_t stage_cap = 0;
    hipEvent_t ev_stage[2] = {nullptr, nullptr}; // behind the device's last read of the staging buffer
    // last batch
    int32_t n_trials = 0;
    double* last_out = nullptr;    // where the records of the last batch live (d_out or the caller's buffer)
    int64_t n_candidates = 0, nnz_candidates = 0;
    double host_log_ms = 0.0;
};

void batch_state_free(BatchState* b) {
    if (!b) return;
    void* ptrs[] = {b->d_q8, b->d_fx, b->d_in, b->d_logpT, b->d_logp_max, b->d_word_off, b->d_c_row, b->d_c_row_ptr, b->d_c_col, b->d_parts, b->d_out,
                    b->d_slab_kept, b->d_slab_nnz, b->d_totals};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    if (b->h_out) (void)hipHostFree(b->h_out);
    for (auto h : b->h_stage) if (h) (void)hipHostFree(h);
    for (auto e : b->ev_stage) if (e) (void)hipEventDestroy(e);
    if (b->h_totals) (void)hipHostFree(b->h_totals);
    for (auto e : b->ev) if (e) (void)hipEventDestroy(e);
    delete b;
}

} // namespace canopy

namespace {

constexpr int kBatchT = 8;       // trials per lane
constexpr int kBatchThreads = 256;
constexpr int kBatchWaves = kBatchThreads / kWave;
constexpr int kInChunk = 16;     // trials staged per host->device copy (two pinned staging buffers in rotation)

int grid_for(int64_t n, int block, int64_t cap = (1 << 20)) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) {                                                                         \
            (void)hipGetLastError();                                                                    \
            return fail((ctx), _e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP,   \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                             \
        }                                                                                               \
    } while (0)

template <typename T>
hipError_t grow(T** p, int64_t* cap, int64_t need) {
    if (need <= *cap && *p) return hipSuccess;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    // a quarter of headroom: hipFree + hipMalloc are synchronous and cost hundreds of microseconds in the middle of a batch,
    // and consecutive batches of one analysis differ by a few per cent in their candidate counts
    const int64_t want = need > 0 ? need + need / 4 + 1 : 1;
    hipError_t e = hipMalloc(p, sizeof(T) * (size_t)want);
    if (e == hipSuccess) *cap = want;
    return e;
}

// ---- layout kernels ---------------------------------------------------------------------------------

// in [n_k][E] (trial-major staging chunk)  ->  out[e * kpad + k0 + k]; 64x64 tiles through LDS
__global__ __launch_bounds__(256) void k_batch_transpose(const double* __restrict__ in, int32_t n_k, int32_t n_events,
                                                         double* __restrict__ out, int32_t kpad, int32_t k0) {
    __shared__ double tile[64][65];
    const int e0 = blockIdx.x * 64, kk0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // 64 x 4
    for (int r = ty; r < 64; r += 4) {
        const int k = kk0 + r, e = e0 + tx;
        tile[r][tx] = (k < n_k && e < n_events) ? in[(size_t)k * n_events + e] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int e = e0 + r, k = kk0 + tx;
        if (e < n_events && k < n_k) out[(size_t)e * kpad + k0 + k] = tile[tx][r];
    }
}

// envelope: logp_max[e] = max over the trials; also clears the padding columns [n_trials, kpad)
__global__ __launch_bounds__(256) void k_batch_colmax(double* __restrict__ logpT, int32_t n_events, int32_t kpad,
                                                      int32_t n_trials, double* __restrict__ logp_max) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_events) return;
    double* row = logpT + (size_t)e * kpad;
    double m = -__builtin_inf();
    for (int k = 0; k < n_trials; ++k) m = row[k] > m ? row[k] : m;
    for (int k = n_trials; k < kpad; ++k) row[k] = 0.0;
    logp_max[e] = m;
}

__global__ __launch_bounds__(256) void k_batch_popc(const uint64_t* __restrict__ mask, int64_t n_words, int64_t n_rows,
                                                    int64_t* __restrict__ cnt) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_words) return;
    uint64_t m = mask[w];
    const int64_t left = n_rows - w * 64;
    if (left < 64) m &= (1ULL << left) - 1ULL;
    cnt[w] = (int64_t)__builtin_popcountll(m);
}

// candidate i (ascending row order) -> its local row index and its length (into c_row_ptr[i + 1])
__global__ __launch_bounds__(256) void k_batch_compact_rows(const uint64_t* __restrict__ mask, int64_t n_words, int64_t n_rows,
                                                            const int64_t* __restrict__ word_incl,
                                                            const int64_t* __restrict__ row_ptr, int64_t* __restrict__ c_row,
                                                            int64_t* __restrict__ c_row_ptr) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w == 0) c_row_ptr[0] = 0;
    if (w >= n_words) return;
    uint64_t m = mask[w];
    const int64_t left = n_rows - w * 64;
    if (left < 64) m &= (1ULL << left) - 1ULL;
    int64_t i = word_incl[w] - (int64_t)__builtin_popcountll(m);
    for (; m != 0ULL; m &= m - 1) {
        const int64_t r = w * 64 + __builtin_ctzll(m);
        c_row[i] = r;
        c_row_ptr[i + 1] = row_ptr[r + 1] - row_ptr[r];
        ++i;
    }
}

__global__ __launch_bounds__(256) void k_batch_compact_cols(const int64_t* __restrict__ c_row, const int64_t* __restrict__ c_row_ptr,
                                                            int64_t n_c, const int64_t* __restrict__ row_ptr,
                                                            const int32_t* __restrict__ col_idx, int32_t* __restrict__ c_col) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_c) return;
    const int64_t src = row_ptr[c_row[i]], dst = c_row_ptr[i], len = c_row_ptr[i + 1] - dst;
    for (int64_t k = 0; k < len; ++k) c_col[dst + k] = col_idx[src + k];
}

// ---- candidates straight from the filter's lists (the fast path at scale) --------------------------------
// The envelope pass only needs the filter of FILTER_REFINE: its per-slab candidate lists already hold, in ascending
// row order, a superset of the rows that survive under the envelope. Three small launches and ONE host read (the two
// totals, to size the compact CSR) replace the keep mask, its 12.5 MB clear, the mask scans and the per-candidate
// length scan: k_batch_mark recomputes each listed row exactly under the envelope (CSR order) and marks the survivors
// in the list (bit 15) while counting them and their entries per slab; k_batch_scan_slabs turns the counts into
// offsets; k_batch_emit writes the compact CSR. One wave per slab: the order stays ascending.
__device__ __forceinline__ double batch_row_sum(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                const double* __restrict__ logp, int64_t r, int* len_out) {
    const int64_t b = row_ptr[r], e = row_ptr[r + 1];
    double s = 0.0;
    for (int64_t k0 = b; k0 < e; k0 += 8) {
        int32_t c[8];
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = (k0 + u < e) ? col_idx[k0 + u] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = logp[c[u]];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + u < e) s = s + v[u]; // plain adds, CSR order
    }
    *len_out = (int)(e - b);
    return s;
}

__global__ __launch_bounds__(256) void k_batch_mark(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                    const double* __restrict__ logp_env, double log_cut, int32_t limit_order,
                                                    uint16_t* __restrict__ cand_list, const unsigned* __restrict__ cand_cnt,
                                                    int slab_rows, int64_t n_slabs, int64_t* __restrict__ slab_kept,
                                                    int64_t* __restrict__ slab_nnz) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t GW = (int64_t)gridDim.x * 4;
    for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < n_slabs; s += GW) {
        const int n = (int)__builtin_amdgcn_readfirstlane(cand_cnt[s]);
        const int64_t base = s * slab_rows;
        long long kept = 0, nnz = 0; // wave-uniform
        for (int j0 = 0; j0 < n; j0 += kWave) {
            const bool on = j0 + lane < n;
            const unsigned entry = on ? (unsigned)cand_list[base + j0 + lane] & 0x7FFFu : 0u;
            int len = 0;
            const double sv = on ? batch_row_sum(row_ptr, col_idx, logp_env, base + (int64_t)entry, &len) : 0.0;
            const bool keep = on && (sv >= log_cut) && (limit_order == 0 || len <= limit_order);
            if (on) cand_list[base + j0 + lane] = (uint16_t)(entry | (keep ? 0x8000u : 0u));
            kept += __builtin_popcountll(__ballot(keep));
            int l = keep ? len : 0;
#pragma unroll
            for (int d = kWave / 2; d >= 1; d >>= 1) l += __shfl_xor(l, d, kWave);
            nnz += l;
        }
        if (lane == 0) { slab_kept[s] = kept; slab_nnz[s] = nnz; }
    }
}

// exclusive prefix of the two per-slab count arrays, in place, by ONE workgroup; totals -> totals[0], totals[1]
__global__ __launch_bounds__(1024) void k_batch_scan_slabs(int64_t* __restrict__ kept, int64_t* __restrict__ nnz, int64_t n,
                                                           int64_t* __restrict__ totals) {
    __shared__ long long s_a[1024], s_b[1024];
    const int tid = threadIdx.x;
    const int64_t per = (n + 1023) / 1024, lo = (int64_t)tid * per, hi = lo + per < n ? lo + per : n;
    long long a = 0, b = 0;
    for (int64_t i = lo; i < hi; ++i) { a += kept[i]; b += nnz[i]; }
    s_a[tid] = a; s_b[tid] = b;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { // Hillis-Steele inclusive scan of the 1024 thread sums
        const long long xa = tid >= d ? s_a[tid - d] : 0, xb = tid >= d ? s_b[tid - d] : 0;
        __syncthreads();
        s_a[tid] += xa; s_b[tid] += xb;
        __syncthreads();
    }
    long long ra = s_a[tid] - a, rb = s_b[tid] - b; // exclusive prefix of this thread's range
    for (int64_t i = lo; i < hi; ++i) {
        const long long ka = kept[i], kb = nnz[i];
        kept[i] = ra; nnz[i] = rb;
        ra += ka; rb += kb;
    }
    if (tid == 1023) { totals[0] = s_a[1023]; totals[1] = s_b[1023]; }
}

__global__ __launch_bounds__(256) void k_batch_emit(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ col_idx,
                                                    const uint16_t* __restrict__ cand_list, const unsigned* __restrict__ cand_cnt,
                                                    int slab_rows, int64_t n_slabs, const int64_t* __restrict__ slab_row_off,
                                                    const int64_t* __restrict__ slab_nnz_off, int64_t* __restrict__ c_row,
                                                    int64_t* __restrict__ c_row_ptr, int32_t* __restrict__ c_col) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t GW = (int64_t)gridDim.x * 4;
    if (blockIdx.x == 0 && threadIdx.x == 0) c_row_ptr[0] = 0;
    for (int64_t s = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); s < n_slabs; s += GW) {
        const int n = (int)__builtin_amdgcn_readfirstlane(cand_cnt[s]);
        const int64_t base = s * slab_rows;
        int64_t pos = slab_row_off[s], off = slab_nnz_off[s]; // wave-uniform running positions
        for (int j0 = 0; j0 < n; j0 += kWave) {
            const bool on = j0 + lane < n;
            const unsigned entry = on ? (unsigned)cand_list[base + j0 + lane] : 0u;
            const bool keep = on && (entry & 0x8000u) != 0;
            const int64_t r = base + (int64_t)(entry & 0x7FFFu);
            const int64_t b = keep ? row_ptr[r] : 0;
            const int len = keep ? (int)(row_ptr[r + 1] - b) : 0;
            const unsigned long long m = __ballot(keep);
            int incl = len; // inclusive prefix of the kept lengths in lane order
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int o = __shfl_up(incl, d, kWave);
                if (lane >= d) incl += o;
            }
            if (keep) {
                const int64_t i = pos + (int64_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const int64_t dst = off + incl - len;
                c_row[i] = r;
                c_row_ptr[i + 1] = off + incl;
                for (int k = 0; k < len; ++k) c_col[dst + k] = col_idx[b + k];
            }
            pos += __builtin_popcountll(m);
            off += __shfl(incl, kWave - 1, kWave);
        }
    }
}

// ---- the K-trial row kernel ---------------------------------------------------------------------------
struct BatchArgs {
    const int64_t* c_row;
    const int64_t* c_row_ptr;
    const int32_t* c_col;
    const double* logpT; // [E][kpad]
    int64_t n_c;
    int64_t row_offset;
    int32_t kpad;
    int32_t limit_order;
    double log_cut;
    double* parts; // [gridDim.x][kpad][8]
};

__global__ __launch_bounds__(kBatchThreads) void k_batch_rows(BatchArgs a) {
    __shared__ double s_red[kBatchWaves][kBatchT][kAccDoubles];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const int k0 = blockIdx.y * kBatchT;
    Acc acc[kBatchT];
#pragma unroll
    for (int t = 0; t < kBatchT; ++t) acc_init(acc[t]);
    const int64_t stride = (int64_t)gridDim.x * kBatchThreads;
    const int64_t n_pad = (a.n_c + kWave - 1) / kWave * kWave; // whole waves stay together (ballots inside)
    for (int64_t i = (int64_t)blockIdx.x * kBatchThreads + tid; i < n_pad; i += stride) {
        const bool valid = i < a.n_c;
        const int64_t b = valid ? a.c_row_ptr[i] : 0, e = valid ? a.c_row_ptr[i + 1] : 0;
        double s[kBatchT];
#pragma unroll
        for (int t = 0; t < kBatchT; ++t) s[t] = 0.0;
        for (int64_t k = b; k < e; k += 2) { // two entries per step: 16 independent 8-byte values in flight
            const bool two = k + 1 < e;
            const int32_t c0 = a.c_col[k];
            const int32_t c1 = two ? a.c_col[k + 1] : c0;
            const double4* p0 = reinterpret_cast<const double4*>(a.logpT + (size_t)c0 * a.kpad + k0);
            const double4* p1 = reinterpret_cast<const double4*>(a.logpT + (size_t)c1 * a.kpad + k0);
            const double4 v00 = p0[0], v01 = p0[1], v10 = p1[0], v11 = p1[1];
            const double x0[kBatchT] = {v00.x, v00.y, v00.z, v00.w, v01.x, v01.y, v01.z, v01.w};
            const double x1[kBatchT] = {v10.x, v10.y, v10.z, v10.w, v11.x, v11.y, v11.z, v11.w};
#pragma unroll
            for (int t = 0; t < kBatchT; ++t) s[t] = s[t] + x0[t]; // plain adds, CSR order
            if (two) {
#pragma unroll
                for (int t = 0; t < kBatchT; ++t) s[t] = s[t] + x1[t];
            }
        }
        const int len = (int)(e - b);
        const double grow = valid ? (double)(a.row_offset + a.c_row[i]) : -1.0;
#pragma unroll
        for (int t = 0; t < kBatchT; ++t) row_epilogue(acc[t], s[t], len, valid, a.log_cut, a.limit_order, grow);
    }
#pragma unroll
    for (int t = 0; t < kBatchT; ++t) {
        acc_wave_reduce(acc[t]);
        if (lane == 0) acc_store(&s_red[wave][t][0], acc[t]);
    }
    __syncthreads();
    if (tid < kBatchT) {
        Acc r;
        acc_load(r, &s_red[0][tid][0]);
        for (int w = 1; w < kBatchWaves; ++w) {
            Acc o;
            acc_load(o, &s_red[w][tid][0]);
            acc_merge(r, o);
        }
        acc_store(a.parts + ((size_t)blockIdx.x * a.kpad + k0 + tid) * kAccDoubles, r);
    }
}

// one workgroup per trial: fixed thread <- record assignment, fixed tree (as k_finalize)
__global__ __launch_bounds__(256) void k_finalize_batch(const double* __restrict__ parts, int n_blocks, int32_t kpad,
                                                        double* __restrict__ out) {
    __shared__ double s_red[4 * kAccDoubles];
    const int k = blockIdx.x;
    Acc acc;
    acc_init(acc);
    for (int i = threadIdx.x; i < n_blocks; i += 256) {
        Acc o;
        acc_load(o, parts + ((size_t)i * kpad + k) * kAccDoubles);
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
        double* o = out + (size_t)k * kAccDoubles;
        o[0] = rh; o[1] = rl; o[2] = lh; o[3] = ll;
        o[4] = t.n_kept; o[5] = t.n_certain; o[6] = t.max_logp; o[7] = t.argmax;
    }
}

} // namespace

extern "C" {

canopy_status canopy_quantify_batch_async(canopy_ctx* ctx, canopy_model* m, const double* p_trials, int32_t n_trials,
                                          const canopy_quant_params* params, uint32_t batch_flags, void* device_out) try {
    if (!ctx || !m || !p_trials || n_trials < 1) return CANOPY_ERR_INVALID_ARGUMENT;
    if (m->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "model lives on another device");
    canopy_quant_params prm;
    if (params) prm = *params; else canopy_quant_params_default(&prm);
    if (prm.limit_order < 0 || !(prm.cut_off >= 0.0)) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "limit_order < 0 or cut_off < 0");
    if (prm.flags & CANOPY_QUANT_ROW_LOGP)
        return fail(ctx, CANOPY_ERR_UNSUPPORTED, "per-row output is not available for a batch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->batch) {
        ctx->batch = new (std::nothrow) BatchState();
        if (!ctx->batch) return CANOPY_ERR_OUT_OF_MEMORY;
        for (auto& e : ctx->batch->ev) HIP_TRY(ctx, hipEventCreate(&e));
        for (auto& e : ctx->batch->ev_stage) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    BatchState* b = ctx->batch;
    const int32_t E = m->n_events;
    const int32_t kpad = (n_trials + kBatchT - 1) / kBatchT * kBatchT;
    const bool is_log = (batch_flags & CANOPY_BATCH_LOG_INPUT) != 0;

    // ---- 1. host logs (threaded), staged in chunks of trials, transposed on the device to [E][kpad]
    HIP_TRY(ctx, grow(&b->d_logpT, &b->logpT_cap, (int64_t)E * kpad));
    HIP_TRY(ctx, grow(&b->d_logp_max, &b->max_cap, (int64_t)E));
    HIP_TRY(ctx, grow(&b->d_in, &b->in_cap, (int64_t)std::min<int32_t>(kInChunk, n_trials) * E));
    {
        const auto t0 = std::chrono::steady_clock::now();
        const int64_t chunk_doubles = (int64_t)std::min<int32_t>(kInChunk, n_trials) * E;
        if (b->stage_cap < chunk_doubles) {
            for (auto& h : b->h_stage) { if (h) (void)hipHostFree(h); h = nullptr; }
            b->stage_cap = 0;
            for (auto& h : b->h_stage) HIP_TRY(ctx, hipHostMalloc(&h, sizeof(double) * (size_t)(chunk_doubles > 0 ? chunk_doubles : 1)));
            b->stage_cap = chunk_doubles;
        }
        int turn = 0;
        for (int32_t c0 = 0; c0 < n_trials; c0 += kInChunk, turn ^= 1) {
            const int32_t nk = std::min<int32_t>(kInChunk, n_trials - c0);
            const int64_t total = (int64_t)nk * E;
            double* h = b->h_stage[turn];
            HIP_TRY(ctx, hipEventSynchronize(b->ev_stage[turn])); // the copy out of this buffer two chunks ago (a never-recorded event is complete)
            // glibc log on the host (the canonical logp of canopy_model_set_probabilities), on the library's worker threads,
            // while the device copies and transposes the previous chunk
            if (host_log_values(p_trials + (size_t)c0 * E, h, total, is_log) >= 0)
                return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, is_log ? "a log-probability is NaN or above 0" : "a probability is outside [0,1]");
            if (total > 0) {
                HIP_TRY(ctx, hipMemcpyAsync(b->d_in, h, sizeof(double) * (size_t)total, hipMemcpyHostToDevice, ctx->stream));
                HIP_TRY(ctx, hipEventRecord(b->ev_stage[turn], ctx->stream));
                hipLaunchKernelGGL(k_batch_transpose, dim3((E + 63) / 64, (nk + 63) / 64), dim3(256), 0, ctx->stream,
                                   (const double*)b->d_in, nk, E, b->d_logpT, kpad, c0);
                HIP_TRY(ctx, hipGetLastError());
            }
        }
        b->host_log_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (E > 0) {
        hipLaunchKernelGGL(k_batch_colmax, dim3(grid_for(E, 256)), dim3(256), 0, ctx->stream, b->d_logpT, E, kpad, n_trials, b->d_logp_max);
        HIP_TRY(ctx, hipGetLastError());
    }

    // ---- 2. candidate pass: an ordinary quantification under the envelope, keep mask requested
    HIP_TRY(ctx, hipEventRecord(b->ev[0], ctx->stream));
    canopy_quant_params cprm = prm;
    cprm.flags = (prm.flags | CANOPY_QUANT_KEEP_MASK) & ~(uint32_t)CANOPY_QUANT_ROW_LOGP;
    const bool was_timing = ctx->timing_on;
    // the envelope is a probability set of its own (buffers of the batch state, written on the context's stream, which is
    // also the stream the un-pipelined candidate pass runs on): the model's own sets are neither read nor changed
    HIP_TRY(ctx, grow(&b->d_q8, &b->q8_cap, (((int64_t)E + 15) / 16) * 16 + 16));
    HIP_TRY(ctx, grow(&b->d_fx, &b->fx_cap, (int64_t)(E > 0 ? E : 1)));
    PSet envelope;
    envelope.external = true;
    envelope.d_logp = b->d_logp_max;
    envelope.d_q8 = b->d_q8;
    envelope.d_fx = b->d_fx;
    envelope.write_stream = ctx->stream;
    {
        std::lock_guard<std::mutex> lk(m->mu);
        envelope.version = ++m->logp_version; // (AUTO's feedback is keyed by it: never that of one of the model's sets)
    }
    ctx->timing_on = false;
    ctx->filter_only_done = false;
    PassOptions popt;
    popt.filter_only = true; // at scale AUTO runs FILTER_REFINE: only its filter is needed (lists instead of a mask)
    canopy_status st = enqueue_pass(ctx, m, &envelope, &cprm, popt);
    const bool fast = st == CANOPY_OK && ctx->filter_only_done;
    canopy_partial env;
    memset(&env, 0, sizeof(env));
    if (st == CANOPY_OK && !fast) st = canopy_quantify_fetch(ctx, &env);
    int64_t n_c = (int64_t)env.n_kept, nnz_c = 0;
    if (fast) {
        // ---- 3a. mark the exact survivors in the lists, count per slab, scan, ONE host read of the two totals
        canopy_ctx::PassScratch& S = ctx->sets[0];
        const int64_t n_slabs = ctx->filter_only_slabs;
        const int slab_rows = ctx->filter_only_slab_rows;
        hipError_t e = grow(&b->d_slab_kept, &b->slab_cap, n_slabs);
        if (e == hipSuccess) e = grow(&b->d_slab_nnz, &b->slab_cap2, n_slabs);
        if (e == hipSuccess && !b->d_totals) e = hipMalloc(&b->d_totals, 2 * sizeof(int64_t));
        if (e == hipSuccess && !b->h_totals) e = hipHostMalloc(&b->h_totals, 2 * sizeof(int64_t));
        const double log_cut = prm.cut_off > 0.0 ? log(prm.cut_off) : -INFINITY;
        const int g = grid_for(n_slabs, 4, 4096);
        if (e == hipSuccess) e = hipMemsetAsync(S.d_fr_count, 0, 2 * sizeof(unsigned long long), ctx->stream); // (the filter's total: no final reduce took it)
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_batch_mark, dim3(g), dim3(256), 0, ctx->stream, (const int64_t*)m->d_row_ptr, (const int32_t*)m->d_col_idx,
                               (const double*)b->d_logp_max, log_cut, prm.limit_order, S.d_fr_list, (const unsigned*)S.d_fr_cnt, slab_rows,
                               n_slabs, b->d_slab_kept, b->d_slab_nnz);
            hipLaunchKernelGGL(k_batch_scan_slabs, dim3(1), dim3(1024), 0, ctx->stream, b->d_slab_kept, b->d_slab_nnz, n_slabs, b->d_totals);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(b->h_totals, b->d_totals, 2 * sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) { n_c = b->h_totals[0]; nnz_c = b->h_totals[1]; }
        if (e == hipSuccess) e = grow(&b->d_c_row_ptr, &b->c_cap, n_c + 1);
        if (e == hipSuccess) e = grow(&b->d_c_row, &b->c_row_cap, n_c);
        if (e == hipSuccess) e = grow(&b->d_c_col, &b->col_cap, nnz_c);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_batch_emit, dim3(g), dim3(256), 0, ctx->stream, (const int64_t*)m->d_row_ptr, (const int32_t*)m->d_col_idx,
                               (const uint16_t*)S.d_fr_list, (const unsigned*)S.d_fr_cnt, slab_rows, n_slabs, (const int64_t*)b->d_slab_kept,
                               (const int64_t*)b->d_slab_nnz, b->d_c_row, b->d_c_row_ptr, b->d_c_col);
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            (void)hipGetLastError();
            st = fail(ctx, e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP, std::string("batch candidates: ") + hipGetErrorString(e));
        }
    }
    ctx->timing_on = was_timing;
    if (st != CANOPY_OK) return st;
    const int64_t n_words = (m->n_rows + 63) / 64;

    // ---- 3b. (other kernel variants) compact the candidates out of the keep mask
    if (!fast) {
    HIP_TRY(ctx, grow(&b->d_c_row_ptr, &b->c_cap, n_c + 1));
    HIP_TRY(ctx, grow(&b->d_c_row, &b->c_row_cap, n_c));
    if (n_c > 0) {
        HIP_TRY(ctx, grow(&b->d_word_off, &b->word_cap, n_words));
        hipLaunchKernelGGL(k_batch_popc, dim3(grid_for(n_words, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t*)ctx->d_keep_mask, n_words, m->n_rows, b->d_word_off);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, scan_inclusive_i64(b->d_word_off, n_words, ctx->stream));
        hipLaunchKernelGGL(k_batch_compact_rows, dim3(grid_for(n_words, 256)), dim3(256), 0, ctx->stream,
                           (const uint64_t*)ctx->d_keep_mask, n_words, m->n_rows, (const int64_t*)b->d_word_off,
                           (const int64_t*)m->d_row_ptr, b->d_c_row, b->d_c_row_ptr);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, scan_inclusive_i64(b->d_c_row_ptr + 1, n_c, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&nnz_c, b->d_c_row_ptr + n_c, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, grow(&b->d_c_col, &b->col_cap, nnz_c));
        hipLaunchKernelGGL(k_batch_compact_cols, dim3(grid_for(n_c, 256)), dim3(256), 0, ctx->stream,
                           (const int64_t*)b->d_c_row, (const int64_t*)b->d_c_row_ptr, n_c, (const int64_t*)m->d_row_ptr,
                           (const int32_t*)m->d_col_idx, b->d_c_col);
        HIP_TRY(ctx, hipGetLastError());
    } else {
        HIP_TRY(ctx, grow(&b->d_c_col, &b->col_cap, 1));
    }
    } // !fast
    HIP_TRY(ctx, hipEventRecord(b->ev[1], ctx->stream));

    // ---- 4. K trials over the compact CSR, 5. one record per trial
    const int n_chunks = kpad / kBatchT;
    int64_t gx = (8 * (int64_t)ctx->n_cu + n_chunks - 1) / n_chunks; // ~8 workgroups per CU in total
    const int64_t gx_max = (n_c + kBatchThreads - 1) / kBatchThreads;
    if (gx > gx_max) gx = gx_max;
    if (gx < 1) gx = 1;
    HIP_TRY(ctx, grow(&b->d_parts, &b->parts_cap, gx * kpad * kAccDoubles));
    HIP_TRY(ctx, grow(&b->d_out, &b->out_cap, (int64_t)kpad * kAccDoubles));
    BatchArgs a;
    a.c_row = b->d_c_row; a.c_row_ptr = b->d_c_row_ptr; a.c_col = b->d_c_col; a.logpT = b->d_logpT;
    a.n_c = n_c; a.row_offset = m->row_offset; a.kpad = kpad; a.limit_order = prm.limit_order;
    a.log_cut = prm.cut_off > 0.0 ? log(prm.cut_off) : -INFINITY;
    a.parts = b->d_parts;
    hipLaunchKernelGGL(k_batch_rows, dim3((unsigned)gx, (unsigned)n_chunks), dim3(kBatchThreads), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    double* out = device_out ? static_cast<double*>(device_out) : b->d_out;
    hipLaunchKernelGGL(k_finalize_batch, dim3((unsigned)n_trials), dim3(256), 0, ctx->stream, (const double*)b->d_parts, (int)gx, kpad, out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(b->ev[2], ctx->stream));
    b->n_trials = n_trials;
    b->last_out = out;
    b->n_candidates = n_c;
    b->nnz_candidates = nnz_c;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_quantify_batch_fetch(canopy_ctx* ctx, canopy_partial* partials, int32_t n_trials, canopy_batch_info* info) try {
    if (!ctx || !partials) return CANOPY_ERR_INVALID_ARGUMENT;
    BatchState* b = ctx->batch;
    if (!b || b->n_trials == 0) return fail(ctx, CANOPY_ERR_STATE, "no batch has been enqueued");
    if (n_trials != b->n_trials) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "n_trials differs from the enqueued batch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t n = (int64_t)n_trials * kAccDoubles;
    if (b->h_out_cap < n) {
        if (b->h_out) (void)hipHostFree(b->h_out);
        b->h_out = nullptr; b->h_out_cap = 0;
        HIP_TRY(ctx, hipHostMalloc(&b->h_out, sizeof(double) * (size_t)n));
        b->h_out_cap = n;
    }
    HIP_TRY(ctx, hipMemcpyAsync(b->h_out, b->last_out, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(partials, b->h_out, sizeof(double) * (size_t)n);
    if (info) {
        float c = 0.f, t = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&c, b->ev[0], b->ev[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&t, b->ev[1], b->ev[2]));
        info->n_candidates = b->n_candidates;
        info->nnz_candidates = b->nnz_candidates;
        info->candidate_ms = c;
        info->trials_ms = t;
        info->host_log_ms = b->host_log_ms;
    }
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_quantify_batch(canopy_ctx* ctx, canopy_model* m, const double* p_trials, int32_t n_trials,
                                    const canopy_quant_params* params, uint32_t batch_flags, canopy_quant_result* results,
                                    canopy_batch_info* info) try {
    if (!results) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_status st = canopy_quantify_batch_async(ctx, m, p_trials, n_trials, params, batch_flags, nullptr);
    if (st != CANOPY_OK) return st;
    std::vector<canopy_partial> parts((size_t)n_trials);
    canopy_batch_info bi;
    st = canopy_quantify_batch_fetch(ctx, parts.data(), n_trials, &bi);
    if (st != CANOPY_OK) return st;
    for (int32_t k = 0; k < n_trials; ++k) {
        st = canopy_combine_partials(&parts[(size_t)k], 1, params ? params->flags : 0, &results[k]);
        if (st != CANOPY_OK) return st;
        results[k].elapsed_kernel_ms = (bi.candidate_ms + bi.trials_ms) / n_trials;
    }
    if (info) *info = bi;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

} // extern "C"

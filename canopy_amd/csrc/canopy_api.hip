// canopy_api.hip -- the C ABI of include/canopy.h over the gfx950 kernels.
// Host code only; every entry point returns a canopy_status and is a function-try-block closed by
// CANOPY_CATCH_ALL (canopy_guard.h): the reference's error style is C++ exceptions,
// src/mef/openpsa/error.h:23-96, which cannot cross a C boundary. There is deliberately NO CPU fallback anywhere in this file.
#include <math.h>
#include <string.h>

#include <atomic>
#include <new>

#include "canopy_device.h"
#include "canopy_internal.h"

using namespace canopy;

namespace {

uint64_t next_model_id() { // contexts may be used from different host threads
    static std::atomic<uint64_t> counter{0};
    return ++counter;
}

canopy_status fail(canopy_ctx* ctx, canopy_status st, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return st;
}

#define HIP_TRY(ctx, expr)                                                                     \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) {                                                                \
            (void)hipGetLastError();                                                           \
            return fail((ctx), _e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP, \
                        std::string(#expr) + ": " + hipGetErrorString(_e));                   \
        }                                                                                      \
    } while (0)

struct CSumH {
    double s = 0.0, c = 0.0;
    void add(double x) { // Neumaier
        const double t = s + x;
        if (fabs(s) >= fabs(x)) c += (s - t) + x; else c += (x - t) + s;
        s = t;
    }
};

canopy_status ensure_block_partials(canopy_ctx* ctx, canopy_ctx::PassScratch& S, int grid) {
    if (grid <= S.block_partials_cap) return CANOPY_OK;
    if (S.d_block_partials) (void)hipFree(S.d_block_partials);
    S.d_block_partials = nullptr;
    S.block_partials_cap = 0;
    HIP_TRY(ctx, hipMalloc(&S.d_block_partials, sizeof(double) * kAccDoubles * (size_t)grid));
    S.block_partials_cap = grid;
    if (&S == &ctx->sets[0]) ++ctx->scratch_gen; // captured passes (always lane 0) hold the old pointer
    return CANOPY_OK;
}

// Orders the context's stream behind the side stream's work (device-side wait, no host blocking).
canopy_status join_side(canopy_ctx* ctx) {
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    return CANOPY_OK;
}

void free_model_buffers(canopy_model* m) {
    if (m->d_row_ptr) (void)hipFree(m->d_row_ptr);
    if (m->d_col_idx) (void)hipFree(m->d_col_idx);
    if (m->d_logp) (void)hipFree(m->d_logp);
    if (m->d_q8) (void)hipFree(m->d_q8);
    m->d_q8 = nullptr;
    if (m->d_fx) (void)hipFree(m->d_fx);
    m->d_fx = nullptr; m->fx_version = 0;
    if (m->d_ws_stream) (void)hipFree(m->d_ws_stream);
    if (m->d_ws_off) (void)hipFree(m->d_ws_off);
    if (m->d_vs_stream) (void)hipFree(m->d_vs_stream);
    if (m->d_vs_off) (void)hipFree(m->d_vs_off);
    if (m->d_vs_slab_row) (void)hipFree(m->d_vs_slab_row);
    if (m->d_fr_stream) (void)hipFree(m->d_fr_stream);
    if (m->d_fr_off) (void)hipFree(m->d_fr_off);
    m->d_fr_stream = nullptr; m->d_fr_off = nullptr; m->fr_state = 0;
    if (m->fr24.stream) (void)hipFree(m->fr24.stream);
    if (m->fr24.base) (void)hipFree(m->fr24.base);
    if (m->fr24.exc) (void)hipFree(m->fr24.exc);
    m->fr24 = FrP24Owned();
    m->d_vs_stream = nullptr; m->d_vs_off = nullptr; m->d_vs_slab_row = nullptr; m->vs_state = 0;
    m->d_row_ptr = nullptr; m->d_col_idx = nullptr; m->d_logp = nullptr;
    m->d_ws_stream = nullptr; m->d_ws_off = nullptr; m->ws_state = 0;
}

} // namespace

hipError_t canopy::ctx_join_side(canopy_ctx* ctx) {
    for (int k = 0; k < canopy_ctx::kLanes; ++k) { // (lane 0 IS the context's stream: pending only behind a collective)
        canopy_ctx::PassScratch& S = ctx->sets[k];
        if (!S.pending) continue;
        hipError_t e = hipStreamWaitEvent(ctx->stream, S.ev_done, 0);
        if (e != hipSuccess) return e;
        S.pending = false;
    }
    return hipSuccess;
}

extern "C" {

const char* canopy_status_string(canopy_status s) {
    switch (s) {
        case CANOPY_OK: return "ok";
        case CANOPY_ERR_INVALID_ARGUMENT: return "invalid argument";
        case CANOPY_ERR_OUT_OF_MEMORY: return "out of memory";
        case CANOPY_ERR_HIP: return "HIP runtime error";
        case CANOPY_ERR_NO_DEVICE: return "no usable HIP device";
        case CANOPY_ERR_UNSUPPORTED: return "unsupported";
        case CANOPY_ERR_INVALID_MODEL: return "invalid CSR model";
        case CANOPY_ERR_STATE: return "invalid call order";
        case CANOPY_ERR_INTERNAL: return "internal error";
        case CANOPY_ERR_RCCL: return "RCCL error";
    }
    return "unknown status";
}

void canopy_version(int* major, int* minor, int* patch) {
    if (major) *major = CANOPY_VERSION_MAJOR;
    if (minor) *minor = CANOPY_VERSION_MINOR;
    if (patch) *patch = CANOPY_VERSION_PATCH;
}

void canopy_quant_params_default(canopy_quant_params* p) {
    if (!p) return;
    p->approximation = CANOPY_APPROX_BOTH;
    p->limit_order = 20;  // reference settings.h:323
    p->cut_off = 1e-8;    // reference settings.h:332
    p->flags = 0;
    p->kernel = CANOPY_KERNEL_AUTO;
}

// ---- context -----------------------------------------------------------------------------------

canopy_status canopy_ctx_create(const canopy_ctx_params* params, canopy_ctx** out) try {
    if (!out) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) {
        (void)hipGetLastError();
        return CANOPY_ERR_NO_DEVICE; // no CPU path: fail loudly
    }
    const int device = params ? params->device : 0;
    if (device < 0 || device >= n_dev) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_ctx* ctx = new (std::nothrow) canopy_ctx();
    if (!ctx) return CANOPY_ERR_OUT_OF_MEMORY;
    ctx->device = device;
    ctx->n_blocks_override = params ? params->n_blocks : 0;
    canopy_status st = CANOPY_OK;
    do {
        hipDeviceProp_t prop;
        if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) {
            st = CANOPY_ERR_HIP; break;
        }
        ctx->n_cu = prop.multiProcessorCount;
        ctx->lds_per_block_max = prop.sharedMemPerBlock;
        if (params && params->stream) {
            ctx->stream = static_cast<hipStream_t>(params->stream);
        } else {
            if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { st = CANOPY_ERR_HIP; break; }
            ctx->own_stream = true;
        }
        if (hipMalloc(&ctx->d_partial, sizeof(double) * CANOPY_PARTIAL_DOUBLES) != hipSuccess ||
            hipHostMalloc(&ctx->h_partial, sizeof(double) * CANOPY_PARTIAL_DOUBLES) != hipSuccess) {
            st = CANOPY_ERR_OUT_OF_MEMORY; break;
        }
        for (auto& ev : ctx->ev_last)
            if (hipEventCreate(&ev) != hipSuccess) { st = CANOPY_ERR_HIP; break; }
        if (st != CANOPY_OK) break;
        ctx->sets[0].stream = ctx->stream;
        ctx->sets[0].d_partial = ctx->d_partial;
        for (int k = 0; k < canopy_ctx::kLanes && st == CANOPY_OK; ++k) {
            canopy_ctx::PassScratch& S = ctx->sets[k];
            if (k > 0 && (hipStreamCreateWithFlags(&S.stream, hipStreamNonBlocking) != hipSuccess ||
                          hipMalloc(&S.d_partial, sizeof(double) * CANOPY_PARTIAL_DOUBLES) != hipSuccess)) st = CANOPY_ERR_HIP;
            if (st == CANOPY_OK && hipEventCreateWithFlags(&S.ev_done, hipEventDisableTiming) != hipSuccess) st = CANOPY_ERR_HIP;
        }
        if (st != CANOPY_OK) break;
        static const bool no_pipe = CANOPY_TUNE_ENV("CANOPY_NO_PIPELINE") != nullptr;
        ctx->pipeline_on = !no_pipe;
    } while (0);
    if (st != CANOPY_OK) {
        (void)hipGetLastError();
        canopy_ctx_free(ctx);
        return st;
    }
    *out = ctx;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void canopy_ctx_free(canopy_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (int k = 1; k < canopy_ctx::kLanes; ++k) if (ctx->sets[k].stream) (void)hipStreamSynchronize(ctx->sets[k].stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto ev : ctx->ev_last) if (ev) (void)hipEventDestroy(ev);
    for (auto ev : ctx->ev_pool) if (ev) (void)hipEventDestroy(ev);
    for (auto& S : ctx->sets) {
        if (S.d_block_partials) (void)hipFree(S.d_block_partials);
        if (S.d_fr_list) (void)hipFree(S.d_fr_list);
        if (S.d_fr_cnt) (void)hipFree(S.d_fr_cnt);
        if (S.d_fr_count) (void)hipFree(S.d_fr_count);
        if (S.ev_done) (void)hipEventDestroy(S.ev_done);
    }
    for (int k = 1; k < canopy_ctx::kLanes; ++k) {
        if (ctx->sets[k].d_partial) (void)hipFree(ctx->sets[k].d_partial);
        if (ctx->sets[k].stream) (void)hipStreamDestroy(ctx->sets[k].stream);
    }
    if (ctx->d_partial) (void)hipFree(ctx->d_partial);
    if (ctx->h_partial) (void)hipHostFree(ctx->h_partial);
    if (ctx->d_row_logp) (void)hipFree(ctx->d_row_logp);
    if (ctx->d_keep_mask) (void)hipFree(ctx->d_keep_mask);
    if (ctx->d_importance) (void)hipFree(ctx->d_importance);
    if (ctx->h_fr_count) (void)hipHostFree(ctx->h_fr_count);
    if (ctx->d_mc_hits) (void)hipFree(ctx->d_mc_hits);
    if (ctx->h_mc_hits) (void)hipHostFree(ctx->h_mc_hits);
    for (auto ev : ctx->ev_mc) if (ev) (void)hipEventDestroy(ev);
    batch_state_free(ctx->batch);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* canopy_last_error(const canopy_ctx* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

canopy_status canopy_ctx_synchronize(canopy_ctx* ctx) try {
    if (!ctx) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_ctx_set_pipeline(canopy_ctx* ctx, int32_t enable) try {
    if (!ctx) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    ctx->pipeline_on = enable != 0;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void* canopy_ctx_partials_device(canopy_ctx* ctx) { return ctx ? ctx->sets[ctx->last_set].d_partial : nullptr; }

int32_t canopy_last_kernel_variant(const canopy_ctx* ctx) { return ctx ? ctx->last_variant : 0; }

canopy_status canopy_ctx_set_gather_buffer(canopy_ctx* ctx, void* device_buffer, int32_t world_size, int32_t rank) try {
    if (!ctx) return CANOPY_ERR_INVALID_ARGUMENT;
    if (device_buffer && (world_size < 1 || rank < 0 || rank >= world_size))
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "gather buffer: need 0 <= rank < world_size");
    if (ctx->d_gather != static_cast<double*>(device_buffer)) ++ctx->scratch_gen;
    ctx->d_gather = static_cast<double*>(device_buffer);
    for (auto& S : ctx->sets) S.d_gather = nullptr; // (canopy_comm_* gives every lane a buffer of its own afterwards)
    ctx->gather_world = device_buffer ? world_size : 0;
    ctx->gather_rank = device_buffer ? rank : 0;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

// ---- model -------------------------------------------------------------------------------------

canopy_status canopy_model_upload(canopy_ctx* ctx, const canopy_csr_view* csr, uint32_t flags,
                                  canopy_model** out) try {
    if (!ctx || !csr || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (csr->n_rows < 0 || csr->n_events < 0 || !csr->row_ptr) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "bad csr view");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool dev_ptrs = (flags & CANOPY_UPLOAD_DEVICE_POINTERS) != 0;
    const hipMemcpyKind kind = dev_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    int64_t ends[2] = {0, 0};
    if (dev_ptrs) {
        HIP_TRY(ctx, hipMemcpy(&ends[0], csr->row_ptr, sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemcpy(&ends[1], csr->row_ptr + csr->n_rows, sizeof(int64_t), hipMemcpyDeviceToHost));
    } else {
        ends[0] = csr->row_ptr[0];
        ends[1] = csr->row_ptr[csr->n_rows];
    }
    const int64_t nnz = ends[1] - ends[0];
    if (nnz < 0 || (nnz > 0 && !csr->col_idx)) return fail(ctx, CANOPY_ERR_INVALID_MODEL, "row_ptr[R] < row_ptr[0] or col_idx missing");

    canopy_model* m = new (std::nothrow) canopy_model();
    if (!m) return CANOPY_ERR_OUT_OF_MEMORY;
    m->id = next_model_id();
    m->device = ctx->device;
    m->n_rows = csr->n_rows; m->n_events = csr->n_events; m->nnz = nnz; m->row_offset = csr->row_offset;
    canopy_status st = CANOPY_OK;
    int64_t* d_status = nullptr;
    auto cleanup = [&](canopy_status s) {
        if (d_status) (void)hipFree(d_status);
        free_model_buffers(m);
        delete m;
        return s;
    };
#define M_TRY(expr)                                                                             \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess) {                                                                 \
            (void)hipGetLastError();                                                            \
            st = fail(ctx, _e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP, \
                      std::string(#expr) + ": " + hipGetErrorString(_e));                       \
            return cleanup(st);                                                                 \
        }                                                                                       \
    } while (0)
    M_TRY(hipMalloc(&m->d_row_ptr, sizeof(int64_t) * (size_t)(m->n_rows + 1)));
    M_TRY(hipMalloc(&m->d_col_idx, sizeof(int32_t) * (size_t)nnz + 64));
    M_TRY(hipMalloc(&m->d_logp, sizeof(double) * (size_t)(m->n_events > 0 ? m->n_events : 1)));
    M_TRY(hipMemcpyAsync(m->d_row_ptr, csr->row_ptr, sizeof(int64_t) * (size_t)(m->n_rows + 1), kind, ctx->stream));
    if (nnz > 0) M_TRY(hipMemcpyAsync(m->d_col_idx, csr->col_idx, sizeof(int32_t) * (size_t)nnz, kind, ctx->stream));
    M_TRY(hipMemsetAsync(reinterpret_cast<char*>(m->d_col_idx) + sizeof(int32_t) * (size_t)nnz, 0, 64, ctx->stream));
    if (ends[0] != 0) M_TRY(launch_rebase_row_ptr(m->d_row_ptr, m->n_rows + 1, ends[0], ctx->stream));
    {
        // validation also yields max_row_len; it always runs (its verdict is only enforced when
        // the caller asked for it), so a malformed CSR can never index out of bounds later.
        M_TRY(hipMalloc(&d_status, 2 * sizeof(int64_t)));
        const int64_t init[2] = {-1 /* = ~0ULL */, 0};
        M_TRY(hipMemcpyAsync(d_status, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
        M_TRY(launch_validate_csr(m->d_row_ptr, m->d_col_idx, m->n_rows, nnz, m->n_events, d_status, ctx->stream));
        int64_t status[2];
        M_TRY(hipMemcpyAsync(status, d_status, sizeof(status), hipMemcpyDeviceToHost, ctx->stream));
        M_TRY(hipStreamSynchronize(ctx->stream));
        if (status[0] != -1) {
            st = fail(ctx, CANOPY_ERR_INVALID_MODEL,
                      "CSR validation failed at row " + std::to_string(status[0]) +
                          " (row_ptr not monotone, column out of range, or columns not strictly ascending)");
            return cleanup(st);
        }
        m->max_row_len = (int32_t)status[1];
        (void)hipFree(d_status);
        d_status = nullptr;
    }
#undef M_TRY
    (void)flags;
    m->device_bytes = (int64_t)(sizeof(int64_t) * (size_t)(m->n_rows + 1) + sizeof(int32_t) * (size_t)nnz + 64 +
                                sizeof(double) * (size_t)m->n_events);
    *out = m;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_model_synthetic(canopy_ctx* ctx, uint64_t seed, int64_t first_row, int64_t n_rows,
                                     int32_t n_events, int32_t avg_nnz, canopy_model** out) try {
    if (!ctx || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (n_rows < 0 || first_row < 0 || avg_nnz < 2 || avg_nnz > 17 || n_events < 2 * avg_nnz - 2)
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "synthetic: need avg_nnz in [2,17], n_events >= 2*avg_nnz-2");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    canopy_model* m = new (std::nothrow) canopy_model();
    if (!m) return CANOPY_ERR_OUT_OF_MEMORY;
    m->id = next_model_id();
    m->device = ctx->device;
    m->n_rows = n_rows; m->n_events = n_events; m->row_offset = first_row;
    m->max_row_len = 2 * avg_nnz - 2;
    hipError_t e = synth_build_csr(seed, first_row, n_rows, n_events, avg_nnz, &m->d_row_ptr, &m->d_col_idx,
                                   &m->nnz, ctx->stream);
    if (e == hipSuccess) e = hipMalloc(&m->d_logp, sizeof(double) * (size_t)n_events);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_model_buffers(m);
        delete m;
        return fail(ctx, e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP,
                    std::string("synthetic model: ") + hipGetErrorString(e));
    }
    m->device_bytes = (int64_t)(sizeof(int64_t) * (size_t)(n_rows + 1) + sizeof(int32_t) * (size_t)m->nnz + 64 +
                                sizeof(double) * (size_t)n_events);
    *out = m;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void canopy_model_free(canopy_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    free_model_buffers(m);
    delete m;
}

canopy_status canopy_model_get_info(const canopy_model* m, canopy_model_info* info) try {
    if (!m || !info) return CANOPY_ERR_INVALID_ARGUMENT;
    info->n_rows = m->n_rows; info->nnz = m->nnz; info->row_offset = m->row_offset;
    info->n_events = m->n_events; info->max_row_len = m->max_row_len;
    info->device_bytes = m->device_bytes;
    info->stream_bytes = 0;
    if (m->fr_state == 1) // what the quantised filter streams: the 24-bit form when the model has one
        info->stream_bytes = m->fr_total_chunks * (m->fr24.stream ? 768 + 4 : 1024) + (m->n_fr_slabs * fr_parts(m->n_events) + 1) * 8 +
                             m->n_fr_slabs * 8 + (int64_t)m->n_events * 8;
    else if (m->vs_state == 1)
        info->stream_bytes = m->vs_total_chunks * 1024 + (m->n_slabs + 1) * 16 + (m->n_rows + 1) * 8 + (int64_t)m->n_events * 8;
    else if (m->ws_state == 1)
        info->stream_bytes = m->ws_total_chunks * 1024 + (m->n_ws + 1) * 8 + (int64_t)m->n_events * 8;
    info->algorithmic_bytes = (m->n_rows + 1) * 8 + m->nnz * 4 + (int64_t)m->n_events * 8;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_model_download(canopy_ctx* ctx, const canopy_model* m, int64_t* row_ptr, int32_t* col_idx) try {
    if (!ctx || !m) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (row_ptr) HIP_TRY(ctx, hipMemcpyAsync(row_ptr, m->d_row_ptr, sizeof(int64_t) * (size_t)(m->n_rows + 1), hipMemcpyDeviceToHost, ctx->stream));
    if (col_idx && m->nnz > 0) HIP_TRY(ctx, hipMemcpyAsync(col_idx, m->d_col_idx, sizeof(int32_t) * (size_t)m->nnz, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_model_set_probabilities(canopy_ctx* ctx, canopy_model* m, const double* p) try {
    if (!ctx || !m || (!p && m->n_events > 0)) return CANOPY_ERR_INVALID_ARGUMENT;
    for (int32_t j = 0; j < m->n_events; ++j)
        if (!(p[j] >= 0.0 && p[j] <= 1.0))
            return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "probability " + std::to_string(j) + " outside [0,1]");
    std::vector<double> logp((size_t)m->n_events);
    for (int32_t j = 0; j < m->n_events; ++j) logp[(size_t)j] = log(p[j]); // glibc log, once, on the host
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx)); // a pipelined pass may still be reading the old values on the side stream
    if (m->n_events > 0)
        HIP_TRY(ctx, hipMemcpyAsync(m->d_logp, logp.data(), sizeof(double) * (size_t)m->n_events, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // logp is a stack-lifetime buffer
    m->has_p = true;
    ++m->logp_version; // derived tables (the quantised filter table) are stale now
    return CANOPY_OK;
} CANOPY_CATCH_ALL

// ---- derived layouts -----------------------------------------------------------------------------

namespace {

// What AUTO resolves to for this model (before knowing whether the WS build succeeds).
int auto_variant(const canopy_model* m) {
    // measured on MI355X (avg 6 nnz/row, 1e8 rows): the LDS-table row kernel wins up to E = 2048 (two
    // workgroups per CU still fit); beyond that the table costs occupancy and the wave-slab stream wins
    if (m->n_events <= 2048) return CANOPY_KERNEL_ROWS_LDS;
    // round 2, E = 1e5, every row kept: ROWS 64 us / 113 us at 2e6 / 4e6 rows, WAVE_STREAM 72 / 75 (a ~50 us floor: few
    // wave-slabs, one wave each): the stream kernel from 2.5e6 rows
    if (m->n_events <= ws_max_events() && m->n_rows >= (int64_t)2500000) return CANOPY_KERNEL_WAVE_STREAM;
    return CANOPY_KERNEL_ROWS;
}

// Sizes from which the two slab passes replace the row kernel AUTO would run (round 2, pipelined and serial passes at
// E = 1e3 and 1e5: filter/refine is level with the row kernels at ~4e5 .. 7e5 rows serially and ahead from 1e5 rows in a
// pipelined sequence; the fixed-point pass is level with ROWS_LDS at ~3e6 rows, with ROWS / WAVE_STREAM at ~1e6)
int64_t fr_min_rows(int base_variant) { return base_variant == CANOPY_KERNEL_ROWS_LDS ? (int64_t)1000000 : (int64_t)500000; }
int64_t fx_min_rows(int base_variant) { return base_variant == CANOPY_KERNEL_ROWS_LDS ? (int64_t)4000000 : (int64_t)1000000; }

canopy_status build_fr(canopy_ctx* ctx, canopy_model* m) {
    if (m->fr_state != 0) return CANOPY_OK;
    bool ok = false;
    HIP_TRY(ctx, fr_offsets(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, &m->d_fr_off, &m->fr_total_chunks, &m->n_fr_slabs, &ok, ctx->stream));
    if (!ok) { m->fr_state = -1; return CANOPY_OK; }
    HIP_TRY(ctx, fr_build_sorted32(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, m->d_fr_off, m->fr_total_chunks,
                                   m->n_fr_slabs, &m->d_fr_stream, ctx->stream));
    m->device_bytes += (m->fr_total_chunks + 1) * 1024 + (m->n_fr_slabs * fr_parts(m->n_events) + 1) * 8;
    HIP_TRY(ctx, fr_build_pack24(m->d_fr_stream, m->d_fr_off, m->n_fr_slabs, m->fr_total_chunks, m->n_events, &m->fr24, ctx->stream));
    if (m->fr24.stream) {
        m->device_bytes += (m->fr_total_chunks + 1) * (768 + 4) + (m->n_fr_slabs + 1) * 4;
        // the quantised filter reads the 24-bit form only: the 32-bit stream goes, and comes back (ensure_fr32, ~15 ms at
        // 1e8 rows) for the first pass that reads it -- the fixed-point pass, the gathered fp64 filter
        static const bool keep32 = CANOPY_TUNE_ENV("CANOPY_FR_KEEP32") != nullptr; // tuning only
        if (!keep32) {
            (void)hipFree(m->d_fr_stream);
            m->d_fr_stream = nullptr;
            m->device_bytes -= (m->fr_total_chunks + 1) * 1024;
        }
    }
    m->fr_state = 1;
    return CANOPY_OK;
}

// the 32-bit slab stream of a model whose build_fr gave it up for the 24-bit form
canopy_status ensure_fr32(canopy_ctx* ctx, canopy_model* m) {
    if (m->fr_state != 1 || m->d_fr_stream != nullptr) return CANOPY_OK;
    HIP_TRY(ctx, fr_build_sorted32(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, m->d_fr_off, m->fr_total_chunks,
                                   m->n_fr_slabs, &m->d_fr_stream, ctx->stream));
    m->device_bytes += (m->fr_total_chunks + 1) * 1024;
    return CANOPY_OK;
}

canopy_status build_vs(canopy_ctx* ctx, canopy_model* m) {
    if (m->vs_state != 0) return CANOPY_OK;
    bool ok = false;
    HIP_TRY(ctx, vs_build(m->d_row_ptr, m->d_col_idx, m->n_rows, m->nnz, m->n_events, m->max_row_len, &m->d_vs_stream,
                          &m->d_vs_off, &m->d_vs_slab_row, &m->vs_total_chunks, &m->n_slabs, &ok, ctx->stream));
    m->vs_state = ok ? 1 : -1;
    if (ok) m->device_bytes += (m->vs_total_chunks + 1) * 1024 + (m->n_slabs + 1) * 16;
    return CANOPY_OK;
}

canopy_status build_ws(canopy_ctx* ctx, canopy_model* m) {
    if (m->ws_state != 0) return CANOPY_OK;
    if (m->n_events > ws_max_events()) { m->ws_state = -1; return CANOPY_OK; }
    bool ok = false;
    HIP_TRY(ctx, ws_build(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, m->max_row_len, &m->d_ws_stream,
                          &m->d_ws_off, &m->ws_total_chunks, &m->n_ws, &ok, ctx->stream));
    m->ws_state = ok ? 1 : -1;
    if (ok) m->device_bytes += (m->ws_total_chunks + 1) * 1024 + (m->n_ws + 1) * 8;
    return CANOPY_OK;
}

} // namespace

extern "C" canopy_status canopy_model_prepare(canopy_ctx* ctx, canopy_model* m, int32_t kernel) try {
    if (!ctx || !m) return CANOPY_ERR_INVALID_ARGUMENT;
    if (m->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "model lives on another device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int variant = kernel == CANOPY_KERNEL_AUTO ? auto_variant(m) : kernel;
    if (kernel == CANOPY_KERNEL_AUTO && m->n_events >= 1 && m->n_events <= fr_max_events() && m->n_rows >= fr_min_rows(variant)) {
        canopy_status fst = build_fr(ctx, m); // AUTO picks filter/refine for cut_off > 0
        if (fst != CANOPY_OK) return fst;
        // the stream of the exact all-rows kernel (cut_off = 0, per-row output, > 20 % survivors) is built by the
        // first pass that needs it: 0.12 s and 2.45 GB at 1e8 rows that a filter/refine workload never reads
        if (m->fr_state == 1) return CANOPY_OK;
    }
    if (variant == CANOPY_KERNEL_ROWS || variant == CANOPY_KERNEL_ROWS_LDS) return CANOPY_OK;
    if (variant == CANOPY_KERNEL_FILTER_REFINE || variant == CANOPY_KERNEL_FIXED_SUM) { // both read the slab stream
        canopy_status st = build_fr(ctx, m);
        if (st != CANOPY_OK) return st;
        if (m->fr_state != 1) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a filter/refine stream");
        if (variant == CANOPY_KERNEL_FIXED_SUM && m->max_row_len > fx_max_row_len())
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "FIXED_SUM needs rows of at most 63 entries");
        if (variant == CANOPY_KERNEL_FIXED_SUM) return ensure_fr32(ctx, m); // (the filter reads the 24-bit form when there is one)
        return CANOPY_OK;
    }
    if (variant == CANOPY_KERNEL_VALUE_SCATTER) {
        canopy_status st = build_vs(ctx, m);
        if (st != CANOPY_OK) return st;
        if (m->vs_state == 1) return CANOPY_OK;
        if (kernel == CANOPY_KERNEL_VALUE_SCATTER)
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a value-scatter stream");
        return CANOPY_OK; // AUTO: quantify falls back to ROWS
    }
    if (variant != CANOPY_KERNEL_WAVE_STREAM) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "unknown kernel variant");
    canopy_status st = build_ws(ctx, m);
    if (st != CANOPY_OK) return st;
    if (m->ws_state != 1 && kernel == CANOPY_KERNEL_WAVE_STREAM)
        return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a wave-slab stream");
    return CANOPY_OK;
} CANOPY_CATCH_ALL

// ---- quantification ----------------------------------------------------------------------------

canopy_status canopy_quantify_async(canopy_ctx* ctx, const canopy_model* m, const canopy_quant_params* params) try {
    if (!ctx || !m) return CANOPY_ERR_INVALID_ARGUMENT;
    if (m->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "model lives on another device");
    if (!m->has_p) return fail(ctx, CANOPY_ERR_STATE, "canopy_model_set_probabilities has not been called");
    canopy_quant_params dflt;
    if (!params) { canopy_quant_params_default(&dflt); params = &dflt; }
    if (params->limit_order < 0 || !(params->cut_off >= 0.0))
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "limit_order < 0 or cut_off < 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    const bool want_rows_early = (params->flags & CANOPY_QUANT_ROW_LOGP) != 0;
    int variant = params->kernel == CANOPY_KERNEL_AUTO ? auto_variant(m) : params->kernel;
    // AUTO: with a positive cut-off and no per-row output, filter/refine replaces the row kernels once the
    // model is large enough to amortise its launches (fr_min_rows), unless its filter listed too many of
    // the rows last time (fr_leave_at).
    // Where filter/refine stops paying depends on what takes over: against the fixed-point pass (0.69 ms per 1e8 rows
    // whatever the survivors) the crossover is at ~10 % of the rows listed (measured at config 3, pipelined steps, cut-offs
    // 1e-9 .. 1e-13: 0.43 ms at 2 % survivors, 0.55 at 5.7 %, 0.68 at 8 %, 0.69 at 10 %), against the exact CSR-order
    // stream kernel (1.2-1.5 ms) at ~20 %.
    const bool fx_can_take_over = m->n_events >= 1 && m->n_events <= fr_max_events() && m->n_rows >= fx_min_rows(variant) &&
                                  m->max_row_len <= fx_max_row_len() &&
                                  !(ctx->fx_known_id == m->id && ctx->fx_known_fraction > 0.2 &&
                                    ctx->fx_known_cut_off == params->cut_off && ctx->fx_known_version == m->logp_version);
    const double fr_leave_at = fx_can_take_over ? 0.10 : 0.2;
    if (params->kernel == CANOPY_KERNEL_AUTO && params->cut_off > 0.0 && !want_rows_early &&
        m->n_events >= 1 && m->n_events <= fr_max_events() && m->n_rows >= fr_min_rows(variant) &&
        !(ctx->fr_known_id == m->id && ctx->fr_known_fraction > fr_leave_at && ctx->fr_known_cut_off == params->cut_off &&
          ctx->fr_known_version == m->logp_version)) {
        canopy_status pst = build_fr(ctx, const_cast<canopy_model*>(m));
        if (pst != CANOPY_OK) return pst;
        if (m->fr_state == 1) variant = CANOPY_KERNEL_FILTER_REFINE;
    }
    // AUTO, sum-only pass that filter/refine does not serve (cut_off = 0, or a cut-off most rows survive): the
    // order-independent fixed-point pass instead of the exact CSR-order stream kernel, unless IT had to hand > 20 % of
    // the rows to the exact pass last time (huge |s|, clamped entries)
    // (also for E <= 2048 at scale, where the row kernel with the table in LDS would run: 1e8 rows, E = 1e3, cut_off 0:
    // 1.01 ms against 0.54 with the fixed-point table in LDS; 1e7 rows: 0.117 against 0.088)
    if (params->kernel == CANOPY_KERNEL_AUTO && !want_rows_early && fx_can_take_over && variant != CANOPY_KERNEL_FILTER_REFINE) {
        canopy_status pst = build_fr(ctx, const_cast<canopy_model*>(m));
        if (pst != CANOPY_OK) return pst;
        if (m->fr_state == 1) variant = CANOPY_KERNEL_FIXED_SUM;
    }
    if (variant == CANOPY_KERNEL_FILTER_REFINE || variant == CANOPY_KERNEL_FIXED_SUM) {
        if (variant == CANOPY_KERNEL_FILTER_REFINE && !(params->cut_off > 0.0))
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "FILTER_REFINE needs cut_off > 0");
        if (want_rows_early)
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "FILTER_REFINE / FIXED_SUM have no per-row output (bit-exact s_i needs the CSR order)");
        if (variant == CANOPY_KERNEL_FIXED_SUM && m->max_row_len > fx_max_row_len())
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "FIXED_SUM needs rows of at most 63 entries");
        canopy_status pst = build_fr(ctx, const_cast<canopy_model*>(m));
        if (pst != CANOPY_OK) return pst;
        if (m->fr_state != 1) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a filter/refine stream");
        // the 32-bit form of the stream, if this pass reads it and the model gave it up for the 24-bit one
        const bool reads_p24 = variant == CANOPY_KERNEL_FILTER_REFINE && fr_filter_is_quantised(m->n_events) && m->fr24.stream != nullptr;
        if (!reads_p24 && m->d_fr_stream == nullptr) {
            if (ctx->capturing) return fail(ctx, CANOPY_ERR_STATE, "the slab stream must exist before a capture");
            pst = join_side(ctx);
            if (pst != CANOPY_OK) return pst;
            pst = ensure_fr32(ctx, const_cast<canopy_model*>(m));
            if (pst != CANOPY_OK) return pst;
        }
    }
    if (variant != CANOPY_KERNEL_ROWS && variant != CANOPY_KERNEL_ROWS_LDS && variant != CANOPY_KERNEL_WAVE_STREAM &&
        variant != CANOPY_KERNEL_VALUE_SCATTER && variant != CANOPY_KERNEL_FILTER_REFINE && variant != CANOPY_KERNEL_FIXED_SUM)
        return fail(ctx, CANOPY_ERR_UNSUPPORTED, "unknown kernel variant");
    if (variant == CANOPY_KERNEL_VALUE_SCATTER) {
        canopy_status pst = build_vs(ctx, const_cast<canopy_model*>(m)); // lazy, one-off, independent of p
        if (pst != CANOPY_OK) return pst;
        if (m->vs_state != 1) {
            if (params->kernel == CANOPY_KERNEL_VALUE_SCATTER)
                return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a value-scatter stream");
            variant = CANOPY_KERNEL_ROWS;
        }
    }
    if (variant == CANOPY_KERNEL_ROWS_LDS && (size_t)m->n_events > rows_lds_table_limit_events())
        return fail(ctx, CANOPY_ERR_UNSUPPORTED, "logp table does not fit the LDS budget of ROWS_LDS");
    if (variant == CANOPY_KERNEL_WAVE_STREAM) {
        canopy_status pst = build_ws(ctx, const_cast<canopy_model*>(m)); // lazy, one-off, independent of p
        if (pst != CANOPY_OK) return pst;
        if (m->ws_state != 1) {
            if (params->kernel == CANOPY_KERNEL_WAVE_STREAM)
                return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a wave-slab stream");
            variant = CANOPY_KERNEL_ROWS;
        }
    }

    const bool want_rows = (params->flags & CANOPY_QUANT_ROW_LOGP) != 0;
    const bool want_imp = (params->flags & CANOPY_QUANT_IMPORTANCE) != 0;
    const bool want_mask = (params->flags & CANOPY_QUANT_KEEP_MASK) != 0 || want_imp; // importance walks the exact keep mask
    if (want_imp && ctx->ext_importance && ctx->ext_importance_n != m->n_events)
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "importance buffer was set for another number of basic events");
    if (want_imp && !ctx->ext_importance && ctx->importance_cap < m->n_events) {
        if (ctx->d_importance) (void)hipFree(ctx->d_importance);
        ctx->d_importance = nullptr; ctx->importance_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_importance, sizeof(double) * (size_t)(m->n_events > 0 ? m->n_events : 1)));
        ctx->importance_cap = m->n_events; ++ctx->scratch_gen;
    }
    if (want_rows && ctx->row_logp_cap < m->n_rows) {
        if (ctx->d_row_logp) (void)hipFree(ctx->d_row_logp);
        ctx->d_row_logp = nullptr; ctx->row_logp_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_row_logp, sizeof(double) * (size_t)(m->n_rows > 0 ? m->n_rows : 1)));
        ctx->row_logp_cap = m->n_rows; ++ctx->scratch_gen;
    }
    const int64_t n_words = (m->n_rows + 63) / 64;
    if (want_mask && ctx->keep_mask_cap < n_words) {
        if (ctx->d_keep_mask) (void)hipFree(ctx->d_keep_mask);
        ctx->d_keep_mask = nullptr; ctx->keep_mask_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_keep_mask, sizeof(uint64_t) * (size_t)(n_words > 0 ? n_words : 1)));
        ctx->keep_mask_cap = n_words; ++ctx->scratch_gen;
    }

    int grid;
    if (variant == CANOPY_KERNEL_WAVE_STREAM)
        grid = ctx->n_blocks_override > 0 ? ctx->n_blocks_override : ws_default_grid(ctx->n_cu, m->n_ws);
    else if (variant == CANOPY_KERNEL_VALUE_SCATTER)
        grid = ctx->n_blocks_override > 0 ? ctx->n_blocks_override : vs_default_grid(ctx->n_cu, m->n_slabs);
    else if (variant == CANOPY_KERNEL_FILTER_REFINE || variant == CANOPY_KERNEL_FIXED_SUM)
        grid = fr_refine_grid(ctx->n_cu);
    else
        grid = ctx->n_blocks_override > 0 ? ctx->n_blocks_override
                                          : rows_kernel_default_grid(ctx->n_cu, variant, m->n_rows);
    const int grid_fx = variant == CANOPY_KERNEL_FIXED_SUM ? fx_sum_grid(ctx->n_cu, m->n_events, m->n_fr_slabs) : 0;
    // Pipelined pass: a sum-only FILTER_REFINE / FIXED_SUM pass takes the next lane (a stream with scratch, record and slot
    // buffer of its own) and runs there from start to end; the device overlaps the lanes. Everything else runs on lane 0,
    // the context's stream, ordered behind whatever the other lanes still hold.
    const bool two_step = variant == CANOPY_KERNEL_FILTER_REFINE || variant == CANOPY_KERNEL_FIXED_SUM;
    const bool pipelined = ctx->pipeline_on && two_step && !want_mask && !want_imp && !want_rows && !ctx->capturing &&
                           !ctx->filter_only_request &&
                           !(ctx->comm && ctx->comm->deferred) && !(ctx->d_gather != nullptr && ctx->comm == nullptr);
    canopy_status st = CANOPY_OK;
    const int set = pipelined ? ctx->pipe_next : 0;
    canopy_ctx::PassScratch& S = ctx->sets[set];
    if (!pipelined) {
        st = join_side(ctx);
        if (st != CANOPY_OK) return st;
    } else if (S.used) {
        // the lane's previous pass (its collective included) must have finished: back-pressure on the HOST (it stays up to
        // two passes ahead of the device), not a wait in a device queue
        HIP_TRY(ctx, hipEventSynchronize(S.ev_done));
        S.pending = false;
    }
    hipStream_t lane = S.stream; // every launch of this pass
    double* const d_gather = S.d_gather ? S.d_gather : ctx->d_gather;
    st = ensure_block_partials(ctx, S, variant == CANOPY_KERNEL_WAVE_STREAM ? grid * 4 : grid + grid_fx);
    if (st != CANOPY_OK) return st;

    RowKernelArgs a;
    a.row_ptr = m->d_row_ptr; a.col_idx = m->d_col_idx; a.logp = m->d_logp;
    a.n_rows = m->n_rows; a.row_offset = m->row_offset; a.n_events = m->n_events;
    a.log_cut = params->cut_off > 0.0 ? log(params->cut_off) : -INFINITY; // host glibc log, as the oracle
    a.limit_order = params->limit_order;
    a.block_partials = S.d_block_partials;
    a.row_logp_out = want_rows ? ctx->d_row_logp : nullptr;
    a.keep_mask_out = want_mask ? ctx->d_keep_mask : nullptr;

    // HIP events around the pass only when somebody will read them (a timing region, or the synchronous canopy_quantify):
    // every record is a marker packet between two kernels, ~5 us of queue time at the 1.25e7-row shard of an 8-GPU run
    hipEvent_t* ev = ctx->ev_last;
    bool timed = ctx->want_last_timing && !ctx->capturing;
    if (ctx->timing_on && !ctx->capturing && ctx->timing_n < ctx->timing_max) {
        ev = &ctx->ev_pool[(size_t)ctx->timing_n * 3];
        ++ctx->timing_n;
        timed = true;
    }
    int n_records = grid;
    if (variant == CANOPY_KERNEL_VALUE_SCATTER) {
        if (want_mask) // slabs start at arbitrary rows: the kernel ORs partial words into a cleared mask
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_keep_mask, 0, sizeof(uint64_t) * (size_t)(n_words > 0 ? n_words : 1), ctx->stream));
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ev[0], lane));
    if (variant == CANOPY_KERNEL_FILTER_REFINE || variant == CANOPY_KERNEL_FIXED_SUM) {
        // delta covers the rounding difference between the unordered pass-A sum and the CSR-order sum
        const double cand_cut = a.log_cut - (1e-9 + 1e-11 * (double)m->max_row_len);
        if (!S.d_fr_count) {
            // {accumulator of the running pass, total of the last finished pass}
            HIP_TRY(ctx, hipMalloc(&S.d_fr_count, 4 * sizeof(unsigned long long)));
            if (!ctx->h_fr_count) HIP_TRY(ctx, hipHostMalloc(&ctx->h_fr_count, sizeof(unsigned long long)));
            // cleared once: a pass's final reduce leaves the accumulator at zero for the next pass on this set
            HIP_TRY(ctx, hipMemsetAsync(S.d_fr_count, 0, 4 * sizeof(unsigned long long), lane));
        }
        ctx->last_set = set;
        ctx->fr_pending_id = m->id;
        ctx->fr_pending_rows = m->n_rows;
        ctx->fr_pending_cut_off = params->cut_off;
        ctx->fr_pending_version = m->logp_version;
        ctx->fr_pending_variant = variant;
        const int slab_rows = fr_slab_rows(m->n_events);
        const int64_t list_len = m->n_fr_slabs * slab_rows;
        if (S.fr_list_cap < list_len || !S.d_fr_list) {
            if (S.d_fr_list) (void)hipFree(S.d_fr_list);
            S.d_fr_list = nullptr; S.fr_list_cap = 0;
            HIP_TRY(ctx, hipMalloc(&S.d_fr_list, sizeof(uint16_t) * (size_t)(list_len > 0 ? list_len : 1)));
            S.fr_list_cap = list_len; if (set == 0) ++ctx->scratch_gen;
        }
        const int64_t cnt_need = m->n_fr_slabs * 8; // the fixed-point pass keeps up to eight list units per slab
        if (S.fr_cnt_cap < cnt_need || !S.d_fr_cnt) {
            if (S.d_fr_cnt) (void)hipFree(S.d_fr_cnt);
            S.d_fr_cnt = nullptr; S.fr_cnt_cap = 0;
            HIP_TRY(ctx, hipMalloc(&S.d_fr_cnt, sizeof(unsigned) * (size_t)(cnt_need > 0 ? cnt_need : 1)));
            S.fr_cnt_cap = cnt_need; if (set == 0) ++ctx->scratch_gen;
        }
        if (want_mask && variant == CANOPY_KERNEL_FILTER_REFINE && !ctx->filter_only_request) // the refine pass ORs the exact keep bits into a cleared mask
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_keep_mask, 0, sizeof(uint64_t) * (size_t)(n_words > 0 ? n_words : 1), ctx->stream));
        bool q8_overlapped = false; // the quantised filter ran with adding / finishing waves: lists in 4096-row units
        if (variant == CANOPY_KERNEL_FIXED_SUM) {
            canopy_model* mm = const_cast<canopy_model*>(m);
            if (!mm->d_fx) HIP_TRY(ctx, hipMalloc(&mm->d_fx, sizeof(uint64_t) * (size_t)(m->n_events > 0 ? m->n_events : 1)));
            if (ctx->capturing || mm->fx_version != m->logp_version || mm->fx_version == 0) { // (a captured pass always rebuilds it)
                if (!ctx->capturing) { // every lane reads the table: rebuilt behind all of them, and finished before any goes on
                    st = join_side(ctx);
                    if (st != CANOPY_OK) return st;
                }
                HIP_TRY(ctx, launch_fx_table(m->d_logp, m->n_events, mm->d_fx, ctx->stream));
                if (!ctx->capturing) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                mm->fx_version = m->logp_version;
            }
            // pass A writes records [0, grid_fx) and every word of the keep mask; the exact pass adds records [grid_fx, ..)
            HIP_TRY(ctx, launch_fx_sum(m->d_fr_stream, m->d_fr_off, m->fr_total_chunks, mm->d_fx, m->d_row_ptr, m->n_rows, m->n_events,
                                       m->n_fr_slabs, a.log_cut, params->limit_order, m->max_row_len, S.d_block_partials,
                                       S.d_fr_list, S.d_fr_cnt, S.d_fr_count, want_mask ? ctx->d_keep_mask : nullptr,
                                       grid_fx, lane));
        } else if (fr_filter_is_quantised(m->n_events)) {
            canopy_model* mm = const_cast<canopy_model*>(m);
            if (!mm->d_q8) HIP_TRY(ctx, hipMalloc(&mm->d_q8, (((size_t)m->n_events + 15) / 16) * 16 + 16));
            // the 8-bit table depends on (logp, cut-off) only: rebuilt when either changed
            const bool table_valid = !ctx->capturing && mm->q8_version == m->logp_version && mm->q8_version != 0 &&
                                     mm->q8_log_cut == a.log_cut; // (a captured pass always rebuilds it: p may change between replays)
            if (!table_valid && !ctx->capturing) { // every lane reads the table: rebuilt behind all of them (only when p or
                st = join_side(ctx);               // the cut-off changed), on the pass's own lane, which the host then waits for
                if (st != CANOPY_OK) return st;
                if (lane != ctx->stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            }
            FrP24 p24;
            p24.stream = m->fr24.stream; p24.base = m->fr24.base; p24.exc = m->fr24.exc;
            q8_overlapped = fr_q8_overlapped(&p24, m->n_events, m->max_row_len);
            HIP_TRY(ctx, launch_fr_filter_q8(m->d_fr_stream, &p24, q8_overlapped, m->max_row_len, m->d_fr_off, m->fr_total_chunks, m->d_logp, mm->d_q8, m->n_rows,
                                             m->n_events, m->n_fr_slabs, a.log_cut, S.d_fr_list, S.d_fr_cnt, S.d_fr_count,
                                             ctx->n_cu, table_valid, lane));
            if (!table_valid && !ctx->capturing) HIP_TRY(ctx, hipStreamSynchronize(lane));
            mm->q8_version = m->logp_version;
            mm->q8_log_cut = a.log_cut;
        } else
        HIP_TRY(ctx, launch_fr_filter(m->d_fr_stream, m->d_fr_off, m->fr_total_chunks, m->d_logp, m->n_rows,
                                      m->n_events, m->n_fr_slabs, cand_cut, S.d_fr_list, S.d_fr_cnt, S.d_fr_count, ctx->n_cu,
                                      lane));
        int unit_rows = slab_rows, units_per_slab = 1; // list units the exact pass walks
        if (variant == CANOPY_KERNEL_FIXED_SUM) fx_list_units(m->n_events, &unit_rows, &units_per_slab);
        else if (q8_overlapped) fr_q8_overlapped_units(&unit_rows, &units_per_slab);
        if (ctx->filter_only_request && variant == CANOPY_KERNEL_FILTER_REFINE) {
            // the batch path lists, marks and compacts the candidates itself (kernels_batch.hip): nothing else to run
            ctx->filter_only_done = true;
            ctx->filter_only_slab_rows = unit_rows;
            ctx->filter_only_slabs = m->n_fr_slabs * units_per_slab;
            ctx->last_variant = variant;
            ctx->last_set = set;
            ctx->fr_pending_id = 0; // its candidate count is consumed by the batch, not by AUTO's feedback
            return CANOPY_OK;
        }
        const int64_t n_units = m->n_fr_slabs * units_per_slab;
        const int64_t g_need = (2 * n_units + 3) / 4 > 0 ? (2 * n_units + 3) / 4 : 1; // two waves per unit, four per workgroup
        if ((int64_t)grid > g_need) grid = (int)g_need; // a small shard does not pay for 2048 workgroups' records
        HIP_TRY(ctx, launch_fr_refine(a, S.d_fr_list, S.d_fr_cnt, unit_rows, n_units,
                                      want_mask ? ctx->d_keep_mask : nullptr, grid, variant == CANOPY_KERNEL_FIXED_SUM, grid_fx,
                                      lane));
        n_records = grid + grid_fx;
    } else if (variant == CANOPY_KERNEL_VALUE_SCATTER) {
        HIP_TRY(ctx, launch_vs_kernel(a, m->d_vs_stream, m->d_vs_off, m->d_vs_slab_row, m->vs_total_chunks, m->n_slabs,
                                      grid, ctx->stream));
    } else if (variant == CANOPY_KERNEL_WAVE_STREAM) {
        const int need_len = (params->limit_order != 0 && params->limit_order < m->max_row_len) ? 1 : 0;
        HIP_TRY(ctx, launch_ws_kernel(a, m->d_ws_stream, m->d_ws_off, m->ws_total_chunks, m->n_ws, need_len, grid,
                                      &n_records, ctx->stream));
    } else {
        HIP_TRY(ctx, launch_rows_kernel(variant, a, grid, ctx->stream));
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ev[1], lane));
    HIP_TRY(ctx, launch_finalize(S.d_block_partials, n_records, S.d_partial, d_gather, ctx->gather_world,
                                 ctx->gather_rank, two_step ? S.d_fr_count : nullptr, lane));
    ctx->last_variant = variant;
    ctx->last_set = set;
    hipStream_t last = lane; // the stream the pass ends on
    if (ctx->comm) { // K5: the one collective of the path, issued here (not by the host language), behind the final reduce
        canopy_status cst = comm_all_reduce_partials(ctx, lane, d_gather, set, timed);
        if (cst != CANOPY_OK) return cst;
        if (comm_stream(ctx)) last = comm_stream(ctx);
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ev[2], last));
    if (pipelined || last != lane) { // someone has to wait for this pass later: the lane's next pass, a fetch, a join
        HIP_TRY(ctx, hipEventRecord(S.ev_done, last));
        S.used = true;
        S.pending = set != 0 || last != lane;
    }
    if (pipelined) ctx->pipe_next = (set + 1) % canopy_ctx::kLanes;
    if (want_imp) { // after the timed pass: one walk over the kept rows
        double* dst = ctx->ext_importance ? ctx->ext_importance : ctx->d_importance;
        HIP_TRY(ctx, hipMemsetAsync(dst, 0, sizeof(double) * (size_t)(m->n_events > 0 ? m->n_events : 1), ctx->stream));
        HIP_TRY(ctx, launch_importance(ctx->d_keep_mask, m->n_rows, m->d_row_ptr, m->d_col_idx, m->d_logp, dst, ctx->stream));
    }
    ctx->last_has_importance = want_imp;
    if (ev == ctx->ev_last) ctx->last_timed = timed;
    ctx->last_has_row_logp = want_rows;
    ctx->last_has_mask = (params->flags & CANOPY_QUANT_KEEP_MASK) != 0;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_quantify_fetch(canopy_ctx* ctx, canopy_partial* out) try {
    if (!ctx || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_partial, ctx->sets[ctx->last_set].d_partial, sizeof(double) * CANOPY_PARTIAL_DOUBLES,
                                hipMemcpyDeviceToHost, ctx->stream));
    if (ctx->fr_pending_id)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_fr_count, ctx->sets[ctx->last_set].d_fr_count + 1, sizeof(unsigned long long),
                                    hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->h_partial, sizeof(canopy_partial));
    if (ctx->fr_pending_id) { // feedback for AUTO: a pass A that lists most rows is not worth its exact pass
        const double frac = ctx->fr_pending_rows > 0 ? (double)*ctx->h_fr_count / (double)ctx->fr_pending_rows : 0.0;
        if (ctx->fr_pending_variant == CANOPY_KERNEL_FIXED_SUM) {
            ctx->fx_known_id = ctx->fr_pending_id;
            ctx->fx_known_cut_off = ctx->fr_pending_cut_off;
            ctx->fx_known_version = ctx->fr_pending_version;
            ctx->fx_known_fraction = frac;
        } else {
            ctx->fr_known_id = ctx->fr_pending_id;
            ctx->fr_known_cut_off = ctx->fr_pending_cut_off;
            ctx->fr_known_version = ctx->fr_pending_version;
            ctx->fr_known_fraction = frac;
        }
        ctx->fr_known_variant = ctx->fr_pending_variant;
        ctx->fr_known_count = (int64_t)*ctx->h_fr_count;
        ctx->fr_known_rows = ctx->fr_pending_rows;
        ctx->fr_pending_id = 0;
    }
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_combine_partials(const canopy_partial* parts, int32_t n, uint32_t flags, canopy_quant_result* r) try {
    if (!parts || n < 0 || !r) return CANOPY_ERR_INVALID_ARGUMENT;
    CSumH rare, l1p;
    double n_kept = 0.0, n_certain = 0.0, max_logp = -INFINITY, argmax = -1.0;
    for (int32_t k = 0; k < n; ++k) { // index (= rank) order: the same on every rank
        rare.add(parts[k].rare_hi); rare.add(parts[k].rare_lo);
        l1p.add(parts[k].log1p_hi); l1p.add(parts[k].log1p_lo);
        n_kept += parts[k].n_kept; n_certain += parts[k].n_certain;
        if (parts[k].n_kept > 0.0 && parts[k].argmax_row >= 0.0 && // a record without kept rows has no argmax
            (parts[k].max_logp > max_logp ||
             (parts[k].max_logp == max_logp && (argmax < 0.0 || parts[k].argmax_row < argmax)))) {
            max_logp = parts[k].max_logp; argmax = parts[k].argmax_row;
        }
    }
    double rare_v = rare.s + rare.c;
    if ((flags & CANOPY_QUANT_CLAMP) && rare_v > 1.0) rare_v = 1.0;
    r->rare_event = rare_v;
    r->mcub = n_certain > 0.0 ? 1.0 : -expm1(l1p.s + l1p.c);
    r->n_kept = (int64_t)n_kept;
    r->argmax_row = (int64_t)argmax;
    r->max_logp = max_logp;
    r->elapsed_kernel_ms = -1.0;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_last_kernel_ms(canopy_ctx* ctx, double* row_kernel_ms, double* total_ms) try {
    if (!ctx) return CANOPY_ERR_INVALID_ARGUMENT;
    if (!ctx->last_timed) return fail(ctx, CANOPY_ERR_STATE, "no timed launch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_last[2]));
    float a = 0.f, b = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->ev_last[0], ctx->ev_last[1]));
    HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->ev_last[0], ctx->ev_last[2]));
    if (row_kernel_ms) *row_kernel_ms = a;
    if (total_ms) *total_ms = b;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_quantify(canopy_ctx* ctx, const canopy_model* m, const canopy_quant_params* params,
                              canopy_quant_result* result) try {
    if (!result) return CANOPY_ERR_INVALID_ARGUMENT;
    const bool was_timing = ctx ? ctx->timing_on : false;
    if (ctx) { ctx->timing_on = false; ctx->want_last_timing = true; } // the synchronous call times itself with ev_last
    canopy_status st = canopy_quantify_async(ctx, m, params);
    if (ctx) { ctx->timing_on = was_timing; ctx->want_last_timing = false; }
    if (st != CANOPY_OK) return st;
    canopy_partial part;
    st = canopy_quantify_fetch(ctx, &part);
    if (st != CANOPY_OK) return st;
    if (ctx->comm && !ctx->comm->deferred) // every rank ran the collective: the combined result, identical on all ranks
        st = canopy_comm_fetch(ctx->comm, params ? params->flags : 0, result, nullptr);
    else
        st = canopy_combine_partials(&part, 1, params ? params->flags : 0, result);
    if (st != CANOPY_OK) return st;
    double k = -1.0, t = -1.0;
    if (canopy_last_kernel_ms(ctx, &k, &t) == CANOPY_OK) result->elapsed_kernel_ms = t;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_fetch_row_logp(canopy_ctx* ctx, const canopy_model* m, int64_t first_row, int64_t n, double* out) try {
    if (!ctx || !m || !out || first_row < 0 || n < 0 || first_row + n > m->n_rows) return CANOPY_ERR_INVALID_ARGUMENT;
    if (!ctx->last_has_row_logp) return fail(ctx, CANOPY_ERR_STATE, "last pass did not request CANOPY_QUANT_ROW_LOGP");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_row_logp + first_row, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_fetch_keep_mask(canopy_ctx* ctx, const canopy_model* m, int64_t first_word, int64_t n_words, uint64_t* out) try {
    if (!ctx || !m || !out || first_word < 0 || n_words < 0 || first_word + n_words > (m->n_rows + 63) / 64)
        return CANOPY_ERR_INVALID_ARGUMENT;
    if (!ctx->last_has_mask) return fail(ctx, CANOPY_ERR_STATE, "last pass did not request CANOPY_QUANT_KEEP_MASK");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_keep_mask + first_word, sizeof(uint64_t) * (size_t)n_words, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_fetch_importance(canopy_ctx* ctx, const canopy_model* m, double* contrib) try {
    if (!ctx || !m || (!contrib && m->n_events > 0)) return CANOPY_ERR_INVALID_ARGUMENT;
    if (!ctx->last_has_importance) return fail(ctx, CANOPY_ERR_STATE, "last pass did not request CANOPY_QUANT_IMPORTANCE");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const double* src = ctx->ext_importance ? ctx->ext_importance : ctx->d_importance;
    if (m->n_events > 0)
        HIP_TRY(ctx, hipMemcpyAsync(contrib, src, sizeof(double) * (size_t)m->n_events, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_ctx_set_importance_buffer(canopy_ctx* ctx, void* device_buffer, int32_t n_events) try {
    if (!ctx || (device_buffer && n_events < 0)) return CANOPY_ERR_INVALID_ARGUMENT;
    if (ctx->ext_importance != static_cast<double*>(device_buffer)) ++ctx->scratch_gen;
    ctx->ext_importance = static_cast<double*>(device_buffer);
    ctx->ext_importance_n = device_buffer ? n_events : 0;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

// ---- captured passes ----------------------------------------------------------------------------------------

canopy_status canopy_graph_capture(canopy_ctx* ctx, const canopy_model* m, const canopy_quant_params* params, canopy_graph** out) try {
    if (!ctx || !m || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (ctx->comm) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "captured passes do not carry the cross-GPU collective: detach the communicator");
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    // one ordinary pass first: lazy layouts, scratch buffers and AUTO's choice are settled outside the capture
    canopy_status st = canopy_quantify_async(ctx, m, params);
    if (st != CANOPY_OK) return st;
    HIP_TRY(ctx, canopy::ctx_join_side(ctx)); // (that pass may have been a pipelined one: nothing may be pending when the capture starts)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    canopy_graph* g = new (std::nothrow) canopy_graph();
    if (!g) return CANOPY_ERR_OUT_OF_MEMORY;
    g->device = ctx->device;
    g->model_id = m->id;
    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) { delete g; (void)hipGetLastError(); return fail(ctx, CANOPY_ERR_HIP, std::string("hipStreamBeginCapture: ") + hipGetErrorString(e)); }
    ctx->capturing = true;
    st = canopy_quantify_async(ctx, m, params);
    ctx->capturing = false;
    e = hipStreamEndCapture(ctx->stream, &g->graph);
    if (st == CANOPY_OK && e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
    if (st != CANOPY_OK || e != hipSuccess) {
        (void)hipGetLastError();
        canopy_graph_free(g);
        return st != CANOPY_OK ? st : fail(ctx, CANOPY_ERR_HIP, std::string("graph capture: ") + hipGetErrorString(e));
    }
    g->variant = ctx->last_variant;
    g->scratch_gen = ctx->scratch_gen; // the launches hold raw pointers to the context's scratch as it is now
    g->model = const_cast<canopy_model*>(m);
    g->has_row_logp = ctx->last_has_row_logp; g->has_mask = ctx->last_has_mask; g->has_importance = ctx->last_has_importance;
    *out = g;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_graph_launch(canopy_ctx* ctx, canopy_graph* g) try {
    if (!ctx || !g || !g->exec) return CANOPY_ERR_INVALID_ARGUMENT;
    if (g->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "graph was captured on another device");
    if (g->scratch_gen != ctx->scratch_gen)
        return fail(ctx, CANOPY_ERR_STATE,
                    "the context's scratch or output buffers were reallocated or re-pointed after this graph was captured "
                    "(a pass over a larger model, a first use of a per-row output, a new gather / importance buffer): capture again");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx)); // the captured launches run on the context's stream (lane 0) and may rewrite tables
    HIP_TRY(ctx, hipGraphLaunch(g->exec, ctx->stream));
    ctx->last_set = 0;
    // a captured FILTER_REFINE pass rewrites the model's quantised table with ITS cut-off: the eager cache key is void
    if (g->variant == CANOPY_KERNEL_FILTER_REFINE && g->model) g->model->q8_version = 0;
    ctx->last_variant = g->variant;
    ctx->last_has_row_logp = g->has_row_logp; ctx->last_has_mask = g->has_mask; ctx->last_has_importance = g->has_importance;
    ctx->last_timed = false;
    ctx->fr_pending_id = 0; // the candidate-count feedback belongs to eager passes
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void canopy_graph_free(canopy_graph* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}

// ---- timing region -----------------------------------------------------------------------------

canopy_status canopy_timing_begin(canopy_ctx* ctx, int32_t max_launches) try {
    if (!ctx || max_launches < 1 || max_launches > (1 << 20)) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    while ((int)ctx->ev_pool.size() < max_launches * 3) {
        hipEvent_t ev = nullptr;
        HIP_TRY(ctx, hipEventCreate(&ev));
        ctx->ev_pool.push_back(ev);
    }
    ctx->timing_max = max_launches;
    ctx->timing_n = 0;
    ctx->timing_on = true;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_timing_end(canopy_ctx* ctx, double* row_kernel_ms_sum, double* total_ms_sum, int32_t* n_launches) try {
    if (!ctx) return CANOPY_ERR_INVALID_ARGUMENT;
    if (!ctx->timing_on) return fail(ctx, CANOPY_ERR_STATE, "timing region not open");
    ctx->timing_on = false;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double ks = 0.0, ts = 0.0;
    for (int i = 0; i < ctx->timing_n; ++i) {
        float a = 0.f, b = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->ev_pool[(size_t)i * 3], ctx->ev_pool[(size_t)i * 3 + 1]));
        HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->ev_pool[(size_t)i * 3], ctx->ev_pool[(size_t)i * 3 + 2]));
        ks += a; ts += b;
    }
    if (row_kernel_ms_sum) *row_kernel_ms_sum = ks;
    if (total_ms_sum) *total_ms_sum = ts;
    if (n_launches) *n_launches = ctx->timing_n;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_timing_get(canopy_ctx* ctx, int32_t i, double* row_kernel_ms, double* total_ms) try {
    if (!ctx || i < 0 || i >= ctx->timing_n) return CANOPY_ERR_INVALID_ARGUMENT;
    if (ctx->timing_on) return fail(ctx, CANOPY_ERR_STATE, "timing region still open");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float a = 0.f, b = 0.f;
    HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->ev_pool[(size_t)i * 3], ctx->ev_pool[(size_t)i * 3 + 1]));
    HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->ev_pool[(size_t)i * 3], ctx->ev_pool[(size_t)i * 3 + 2]));
    if (row_kernel_ms) *row_kernel_ms = a;
    if (total_ms) *total_ms = b;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_last_filter_candidates(const canopy_ctx* ctx, int64_t* n_candidates, int64_t* n_rows) {
    if (!ctx || ctx->fr_known_variant == 0) return CANOPY_ERR_STATE;
    if (n_candidates) *n_candidates = ctx->fr_known_count;
    if (n_rows) *n_rows = ctx->fr_known_rows;
    return CANOPY_OK;
}

// ---- host logic --------------------------------------------------------------------------------

canopy_status canopy_synthetic_probabilities(uint64_t seed, int32_t n_events, double* p) try {
    if (n_events < 0 || (!p && n_events > 0)) return CANOPY_ERR_INVALID_ARGUMENT;
    const uint64_t key = stream_key(seed, kStreamProb);
    for (int32_t j = 0; j < n_events; ++j) {
        const double u = (double)(draw_keyed(key, (uint64_t)j, 0) >> 11) * 0x1.0p-53;
        p[j] = pow(10.0, -8.0 + 5.0 * u);
    }
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_synthetic_shard_bounds(canopy_ctx* ctx, uint64_t seed, int64_t n_rows, int32_t avg_nnz,
                                            int32_t n_shards, int64_t* bounds) try {
    if (!ctx || !bounds || n_rows < 0 || n_shards < 1 || avg_nnz < 2 || avg_nnz > 17) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int64_t* d_rp = nullptr;
    HIP_TRY(ctx, synth_row_ptr(seed, 0, n_rows, avg_nnz, &d_rp, ctx->stream)); // lengths + scan only: no columns
    auto at = [&](int64_t r, int64_t* v) { return hipMemcpy(v, d_rp + r, sizeof(int64_t), hipMemcpyDeviceToHost); };
    int64_t nnz = 0;
    hipError_t e = at(n_rows, &nnz);
    bounds[0] = 0;
    for (int32_t k = 1; k < n_shards && e == hipSuccess; ++k) { // the rule of canopy_shard_rows, searched on the device array
        if (nnz == 0) { bounds[k] = n_rows * k / n_shards; continue; }
        const int64_t target = (int64_t)(((__int128)nnz * k) / n_shards);
        int64_t lo = 0, hi = n_rows; // first r with row_ptr[r] >= target
        while (lo < hi && e == hipSuccess) {
            const int64_t mid = lo + (hi - lo) / 2;
            int64_t v = 0;
            e = at(mid, &v);
            if (v >= target) hi = mid; else lo = mid + 1;
        }
        bounds[k] = lo < bounds[k - 1] ? bounds[k - 1] : lo;
    }
    bounds[n_shards] = n_rows;
    (void)hipFree(d_rp);
    HIP_TRY(ctx, e);
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_shard_rows(const int64_t* row_ptr, int64_t n_rows, int32_t n_shards, int64_t* bounds) try {
    if (!row_ptr || !bounds || n_rows < 0 || n_shards < 1) return CANOPY_ERR_INVALID_ARGUMENT;
    const int64_t base = row_ptr[0], nnz = row_ptr[n_rows] - base;
    if (nnz < 0) return CANOPY_ERR_INVALID_MODEL;
    bounds[0] = 0;
    for (int32_t k = 1; k < n_shards; ++k) {
        if (nnz == 0) { bounds[k] = n_rows * k / n_shards; continue; }
        const int64_t target = base + (int64_t)(((__int128)nnz * k) / n_shards);
        int64_t lo = 0, hi = n_rows; // first r with row_ptr[r] >= target
        while (lo < hi) {
            const int64_t mid = lo + (hi - lo) / 2;
            if (row_ptr[mid] >= target) hi = mid; else lo = mid + 1;
        }
        bounds[k] = lo < bounds[k - 1] ? bounds[k - 1] : lo;
    }
    bounds[n_shards] = n_rows;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

} // extern "C"

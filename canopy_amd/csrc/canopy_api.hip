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

void free_model_buffers(canopy_model* m) {
    if (m->d_row_ptr) (void)hipFree(m->d_row_ptr);
    if (m->d_col_idx) (void)hipFree(m->d_col_idx);
    for (auto& P : m->psets) {
        if (P.ev_write) { (void)hipEventSynchronize(P.ev_write); (void)hipEventDestroy(P.ev_write); } // an upload still on its way
        if (P.d_logp) (void)hipFree(P.d_logp);
        if (P.d_q8) (void)hipFree(P.d_q8);
        if (P.d_fx) (void)hipFree(P.d_fx);
        if (P.h_stage) (void)hipHostFree(P.h_stage);
        P = PSet();
    }
    m->cur = -1;
    if (m->d_bs_stream) (void)hipFree(m->d_bs_stream);
    if (m->d_bs_off) (void)hipFree(m->d_bs_off);
    if (m->d_bs_used) (void)hipFree(m->d_bs_used);
    m->d_bs_stream = nullptr; m->d_bs_off = nullptr; m->d_bs_used = nullptr; m->bs_state = 0;
    if (m->d_ws_stream) (void)hipFree(m->d_ws_stream);
    if (m->d_ws_off) (void)hipFree(m->d_ws_off);
    if (m->d_vs_stream) (void)hipFree(m->d_vs_stream);
    if (m->d_vs_off) (void)hipFree(m->d_vs_off);
    if (m->d_vs_slab_row) (void)hipFree(m->d_vs_slab_row);
    if (m->d_fr_stream) (void)hipFree(m->d_fr_stream);
    if (m->d_fr_off) (void)hipFree(m->d_fr_off);
    m->d_fr_stream = nullptr; m->d_fr_off = nullptr; m->fr_state = 0; m->fr_chunks = 0;
    if (m->fr24.stream) (void)hipFree(m->fr24.stream);
    if (m->fr24.base) (void)hipFree(m->fr24.base);
    if (m->fr24.exc) (void)hipFree(m->fr24.exc);
    m->fr24 = FrP24Owned();
    ls_free(&m->ls);
    m->ls_state = 0;
    if (m->d_off16) (void)hipFree(m->d_off16);
    m->d_off16 = nullptr;
    m->d_vs_stream = nullptr; m->d_vs_off = nullptr; m->d_vs_slab_row = nullptr; m->vs_state = 0;
    m->d_row_ptr = nullptr; m->d_col_idx = nullptr;
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
    ctx->id = next_object_id();
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
        if (hipHostMalloc(&ctx->h_ring, sizeof(double) * 2 * CANOPY_PARTIAL_DOUBLES * canopy_ctx::kTicketRing) != hipSuccess) { st = CANOPY_ERR_OUT_OF_MEMORY; break; }
        for (auto& T : ctx->ring)
            if (hipEventCreateWithFlags(&T.ev, hipEventDisableTiming) != hipSuccess) { st = CANOPY_ERR_HIP; break; }
        if (st != CANOPY_OK) break;
        static const bool no_pipe = CANOPY_TUNE_ENV("CANOPY_NO_PIPELINE") != nullptr; // tuning only
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
    if (ctx->comm) { // (the header asks for the communicator to go first) a forgotten one is DETACHED: its handle stays valid for
        canopy_comm* c = ctx->comm; // the caller's canopy_comm_free, which then finds no context to touch
        (void)canopy::ctx_join_side(ctx);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (c->stream) (void)hipStreamSynchronize(c->stream);
        c->ctx = nullptr;
        ctx->comm = nullptr;
    }
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
    for (auto& T : ctx->ring) if (T.ev) (void)hipEventDestroy(T.ev);
    if (ctx->h_ring) (void)hipHostFree(ctx->h_ring);
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
    for (auto& S : ctx->sets) S.pending_lo = 0; // everything enqueued so far has finished (what a writer of a probability set asks)
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_ctx_set_pipeline(canopy_ctx* ctx, int32_t enable) try {
    if (!ctx) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); // nothing in flight while the rotation changes
    for (auto& S : ctx->sets) { S.pending_lo = 0; S.pending = false; }
    ctx->pipeline_on = enable != 0;
    ctx->n_lanes = enable >= 2 ? (enable > canopy_ctx::kLanes ? canopy_ctx::kLanes : enable) : 4;
    if (ctx->pipe_next >= ctx->n_lanes) ctx->pipe_next = 0;
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
    m->id = next_object_id();
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
    m->id = next_object_id();
    m->device = ctx->device;
    m->n_rows = n_rows; m->n_events = n_events; m->row_offset = first_row;
    m->max_row_len = 2 * avg_nnz - 2;
    hipError_t e = synth_build_csr(seed, first_row, n_rows, n_events, avg_nnz, &m->d_row_ptr, &m->d_col_idx,
                                   &m->nnz, ctx->stream);
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
    info->stream_bytes_wide = 0;
    if (m->fr_state == 1 && m->fr_chunks == 1)
        info->stream_bytes_wide = m->fr_total_chunks * 1024 + (m->n_fr_slabs * fr_parts(m->n_events) + 1) * 8 + (int64_t)m->n_events * 8;
    if (m->fr_state == 1 && m->fr_chunks == 1) // the chunks in the form the quantised filter reads: 24-bit when the model has one
        info->stream_bytes = m->fr_total_chunks * (m->fr24.stream ? 768 + 4 : 1024) + (m->n_fr_slabs * fr_parts(m->n_events) + 1) * 8 +
                             m->n_fr_slabs * 8 + (int64_t)m->n_events * 8;
    else if (m->vs_state == 1)
        info->stream_bytes = m->vs_total_chunks * 1024 + (m->n_slabs + 1) * 16 + (m->n_rows + 1) * 8 + (int64_t)m->n_events * 8;
    else if (m->ws_state == 1)
        info->stream_bytes = m->ws_total_chunks * 1024 + (m->n_ws + 1) * 8 + (int64_t)m->n_events * 8;
    info->stream_bytes_rows = m->ls_state == 1 ? m->ls.stream_bytes + (int64_t)m->n_events * 8 : 0;
    info->stream_bytes_exact = 0;
    if (m->bs_state == 1)
        info->stream_bytes_exact = m->bs_used_ss * bs_ss_bytes() + m->n_bs_slabs * 12 + (int64_t)m->n_events * 8;
    else if (m->ws_state == 1)
        info->stream_bytes_exact = m->ws_total_chunks * 1024 + (m->n_ws + 1) * 8 + (int64_t)m->n_events * 8;
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

} // extern "C"

namespace {

// One new version of the model's probabilities. Host side: log (or a copy) of E values into the pinned staging buffer of
// the NEXT set of the model's ring, on the worker threads of host_pool.cpp. Device side: ONE asynchronous copy, enqueued on
// the stream the next pipelined pass will take -- that pass is then ordered behind it by the stream itself, any other lane
// by the set's write event. Passes in flight keep reading the sets they were enqueued with: nothing is drained.
// While a captured pass (hipGraph) of the model exists the current set is rewritten in place instead, behind everything
// in flight: the graph holds that set's pointers.
canopy_status set_values(canopy_ctx* ctx, canopy_model* m, const double* v, bool is_log) {
    if (!ctx || !m || (!v && m->n_events > 0)) return CANOPY_ERR_INVALID_ARGUMENT;
    if (m->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "model lives on another device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t ev = (size_t)(m->n_events > 0 ? m->n_events : 1);
    const bool in_place = m->graph_refs > 0 && m->cur >= 0;
    const int slot = m->cur < 0 ? 0 : (in_place ? m->cur : (m->cur + 1) % kPSets);
    PSet& P = m->psets[slot];
    // the set (device buffer and staging buffer) must have no reader and no copy in flight left
    HIP_TRY(ctx, wait_set_readers(ctx, P));
    if (P.ev_write) HIP_TRY(ctx, hipEventSynchronize(P.ev_write));
    if (!P.d_logp) { // first use of this set: all three resources or none (a half-made set must not look usable to the next call)
        double* d = nullptr;
        double* h = nullptr;
        hipEvent_t evw = nullptr;
        hipError_t e = hipMalloc(&d, sizeof(double) * ev);
        if (e == hipSuccess) e = hipHostMalloc(&h, sizeof(double) * ev);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&evw, hipEventDisableTiming);
        if (e != hipSuccess) {
            if (d) (void)hipFree(d);
            if (h) (void)hipHostFree(h);
            HIP_TRY(ctx, e);
        }
        P.d_logp = d; P.h_stage = h; P.ev_write = evw;
        m->device_bytes += (int64_t)(sizeof(double) * ev);
    }
    const int64_t bad = host_log_values(v, P.h_stage, m->n_events, is_log);
    if (bad >= 0)
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, is_log ? "log-probability " + std::to_string(bad) + " is NaN or above 0"
                                                             : "probability " + std::to_string(bad) + " outside [0,1]");
    hipStream_t st = ctx->stream;
    if (in_place) {
        HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    } else if (ctx->pipeline_on && !ctx->capturing) {
        st = ctx->sets[ctx->pipe_next].stream;
    }
    if (m->n_events > 0)
        HIP_TRY(ctx, hipMemcpyAsync(P.d_logp, P.h_stage, sizeof(double) * (size_t)m->n_events, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(P.ev_write, st));
    {
        std::lock_guard<std::mutex> lk(m->mu);
        P.write_stream = st;
        P.write_id = ++m->write_counter;
        P.version = ++m->logp_version; // derived tables of this set are stale now
        P.q8_valid = false;
        P.fx_valid = false;
        m->cur = slot;
        m->has_p = true;
    }
    return CANOPY_OK;
}

} // namespace

extern "C" {

canopy_status canopy_model_set_probabilities(canopy_ctx* ctx, canopy_model* m, const double* p) try {
    return set_values(ctx, m, p, false);
} CANOPY_CATCH_ALL

canopy_status canopy_model_set_log_probabilities(canopy_ctx* ctx, canopy_model* m, const double* logp) try {
    return set_values(ctx, m, logp, true);
} CANOPY_CATCH_ALL

// ---- derived layouts -----------------------------------------------------------------------------

canopy_status canopy_model_prepare(canopy_ctx* ctx, canopy_model* m, int32_t kernel) try {
    if (!ctx || !m) return CANOPY_ERR_INVALID_ARGUMENT;
    if (m->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "model lives on another device");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int variant = kernel == CANOPY_KERNEL_AUTO ? auto_variant(m) : kernel;
    if (kernel == CANOPY_KERNEL_AUTO && m->n_events >= 1 && m->n_events <= fr_max_events() && m->n_rows >= fr_min_rows(variant, true)) {
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
        if (variant == CANOPY_KERNEL_FIXED_SUM) { // (the filter reads the row stream, or the 24-bit form when there is one)
            st = ensure_fr_chunks(ctx, m);
            if (st != CANOPY_OK) return st;
            return ensure_fr32(ctx, m);
        }
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
    if (variant == CANOPY_KERNEL_BAND_STREAM) {
        canopy_status st = build_bs(ctx, m, kernel == CANOPY_KERNEL_BAND_STREAM);
        if (st != CANOPY_OK) return st;
        if (m->bs_state == 1) return CANOPY_OK;
        if (kernel == CANOPY_KERNEL_BAND_STREAM) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a band stream");
        if (m->n_events > ws_max_events() || m->n_events <= 2048) return CANOPY_OK; // AUTO: quantify falls back to ROWS / ROWS_LDS
        // AUTO: the wave-slab stream instead
    } else if (variant != CANOPY_KERNEL_WAVE_STREAM) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "unknown kernel variant");
    canopy_status st = build_ws(ctx, m);
    if (st != CANOPY_OK) return st;
    if (m->ws_state != 1 && kernel == CANOPY_KERNEL_WAVE_STREAM)
        return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a wave-slab stream");
    return CANOPY_OK;
} CANOPY_CATCH_ALL

// ---- quantification ----------------------------------------------------------------------------

canopy_status canopy_quantify_async(canopy_ctx* ctx, const canopy_model* m, const canopy_quant_params* params) try {
    return canopy_quantify_enqueue(ctx, m, params, nullptr);
} CANOPY_CATCH_ALL

canopy_status canopy_quantify_enqueue(canopy_ctx* ctx, const canopy_model* m, const canopy_quant_params* params,
                                      canopy_ticket* ticket) try {
    if (ticket) *ticket = 0;
    if (!ctx || !m) return CANOPY_ERR_INVALID_ARGUMENT;
    if (m->device != ctx->device) return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "model lives on another device");
    if (!m->has_p || m->cur < 0) return fail(ctx, CANOPY_ERR_STATE, "canopy_model_set_probabilities has not been called");
    canopy_quant_params dflt;
    if (!params) { canopy_quant_params_default(&dflt); params = &dflt; }
    PassOptions opt;
    opt.ticket_out = ticket;
    return enqueue_pass(ctx, m, &m->psets[m->cur], params, opt);
} CANOPY_CATCH_ALL

// The record of a ticketed pass: its lane's final reduce wrote it into pinned host memory, so this is a wait for THAT
// lane's event and a memcpy -- the other lanes keep running, nothing is joined, no device copy is enqueued.
canopy_status canopy_ticket_fetch(canopy_ctx* ctx, canopy_ticket ticket, canopy_partial* out) try {
    if (!ctx || !out || ticket == 0) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_ctx::TicketSlot& T = ctx->ring[ticket % canopy_ctx::kTicketRing];
    if (T.ticket != ticket)
        return fail(ctx, CANOPY_ERR_STATE, "ticket expired or unknown: a record stays fetchable until 64 later passes have been enqueued on its context");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(T.ev));
    canopy_ctx::PassScratch& S = ctx->sets[T.lane];
    if (S.last_seq == ticket) S.pending_lo = 0;
    const double* rec = ctx->h_ring + (size_t)(ticket % canopy_ctx::kTicketRing) * 2 * CANOPY_PARTIAL_DOUBLES;
    memcpy(out, rec, sizeof(canopy_partial));
    unsigned long long count = 0;
    memcpy(&count, rec + CANOPY_PARTIAL_DOUBLES, sizeof(count));
    if (S.last_seq == ticket) apply_feedback(ctx, S, count); // AUTO: a conservative pass that lists most rows is not worth its exact pass
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_ticket_done(canopy_ctx* ctx, canopy_ticket ticket, int32_t* done) try {
    if (!ctx || !done || ticket == 0) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_ctx::TicketSlot& T = ctx->ring[ticket % canopy_ctx::kTicketRing];
    if (T.ticket != ticket) return fail(ctx, CANOPY_ERR_STATE, "ticket expired or unknown");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const hipError_t e = hipEventQuery(T.ev);
    if (e != hipSuccess && e != hipErrorNotReady) HIP_TRY(ctx, e);
    (void)hipGetLastError();
    *done = e == hipSuccess ? 1 : 0;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_quantify_fetch(canopy_ctx* ctx, canopy_partial* out) try {
    if (!ctx || !out) return CANOPY_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, canopy::ctx_join_side(ctx));
    canopy_ctx::PassScratch& S = ctx->sets[ctx->last_set]; // the lane of the last pass: its record, its candidate total
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_partial, S.d_partial, sizeof(double) * CANOPY_PARTIAL_DOUBLES, hipMemcpyDeviceToHost, ctx->stream));
    const bool feedback = S.fb_model != 0 && S.d_fr_count != nullptr && ctx->h_fr_count != nullptr;
    if (feedback)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_fr_count, S.d_fr_count + 1, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& L : ctx->sets) L.pending_lo = 0; // everything enqueued so far has finished
    memcpy(out, ctx->h_partial, sizeof(canopy_partial));
    if (feedback) apply_feedback(ctx, S, *ctx->h_fr_count); // AUTO: a conservative pass that lists most rows is not worth its exact pass
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
    g->model = m;
    { // ... and to the model's CURRENT probability set: from now on canopy_model_set_probabilities rewrites it in place
        std::lock_guard<std::mutex> lk(m->mu);
        ++m->graph_refs;
    }
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
    // a captured FILTER_REFINE pass rewrites the set's quantised table with ITS cut-off: the eager cache key is void
    if (g->variant == CANOPY_KERNEL_FILTER_REFINE && g->model && g->model->cur >= 0) {
        std::lock_guard<std::mutex> lk(g->model->mu);
        g->model->psets[g->model->cur].q8_valid = false;
    }
    for (auto& S : ctx->sets) S.fb_model = 0; // the candidate-count feedback belongs to eager passes
    ctx->last_variant = g->variant;
    ctx->last_has_row_logp = g->has_row_logp; ctx->last_has_mask = g->has_mask; ctx->last_has_importance = g->has_importance;
    ctx->last_timed = false;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void canopy_graph_free(canopy_graph* g) {
    if (!g) return;
    (void)hipSetDevice(g->device);
    if (g->model) { // (the model outlives its graphs: canopy.h)
        std::lock_guard<std::mutex> lk(g->model->mu);
        if (g->model->graph_refs > 0) --g->model->graph_refs;
    }
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

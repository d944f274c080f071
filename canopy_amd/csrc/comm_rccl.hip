// comm_rccl.hip -- K5 behind the C ABI: the ONE collective of the path, issued by the library itself.
//
// Cut-set rows shard trivially (one shard per GPU); the only exchange step is the combine of the per-shard
// partial records. k_finalize writes this rank's 64-byte record into slot `rank` of a world x 8
// buffer and zeroes the other slots; ONE ncclAllReduce(SUM, world * 8 doubles) on the context's stream then
// acts as an all-gather (x + 0 is exact), and every rank folds the records in rank order
// (canopy_combine_partials): bitwise identical results on all ranks, whatever reduction order RCCL picks.
// The message is world * 64 B: latency-bound, xGMI link bandwidth is irrelevant.
//
// Two shapes, both from SURVEY 8(b)/(e):
//   * one process per GPU      canopy_comm_get_unique_id (rank 0) + canopy_comm_init_rank (every rank)
//   * one process, N devices   canopy_comm_init_all + canopy_comm_quantify_all_async (the N collectives of a
//                              pass are fused in ncclGroupStart / ncclGroupEnd)
//
// librccl.so.1 is bound at the first canopy_comm_* call (dlopen): a single-GPU host needs no RCCL at all, and
// a process that already carries an RCCL (a PyTorch-ROCm process) shares that copy instead of loading a second.
#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>

#include <rccl/rccl.h>

#include "canopy_internal.h"

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    std::string error;
};

RcclApi* rccl() { // nullptr-safe singleton; error text kept for canopy_last_error
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        static const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("librccl.so.1 not found: ") + dlerror(); return; }
        bool ok = true;
#define BIND(field, sym)                                                                \
        api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym));      \
        if (!api.field) { ok = false; api.error = std::string("librccl lacks ") + sym; }
        BIND(GetUniqueId, "ncclGetUniqueId")
        BIND(CommInitRank, "ncclCommInitRank")
        BIND(CommInitAll, "ncclCommInitAll")
        BIND(CommDestroy, "ncclCommDestroy")
        BIND(AllReduce, "ncclAllReduce")
        BIND(AllGather, "ncclAllGather")
        BIND(GroupStart, "ncclGroupStart")
        BIND(GroupEnd, "ncclGroupEnd")
        BIND(GetErrorString, "ncclGetErrorString")
        BIND(GetVersion, "ncclGetVersion")
#undef BIND
        if (!ok) { dlclose(api.handle); api.handle = nullptr; }
    });
    return &api;
}

__global__ void k_mc_pack(unsigned long long* dst, const unsigned long long* hits, unsigned long long trials) {
    dst[0] = *hits;
    dst[1] = trials;
}

canopy_status comm_fail(canopy_ctx* ctx, canopy_status st, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    return st;
}

#define RCCL_TRY(ctx, expr)                                                                            \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess)                                                                         \
            return comm_fail((ctx), CANOPY_ERR_RCCL, std::string(#expr) + ": " + rccl()->GetErrorString(_r)); \
    } while (0)

#define HIPC_TRY(ctx, expr)                                                                            \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            (void)hipGetLastError();                                                                   \
            return comm_fail((ctx), _e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP, \
                             std::string(#expr) + ": " + hipGetErrorString(_e));                       \
        }                                                                                              \
    } while (0)

void comm_release(canopy_comm* c) { // device / host resources only (not the RCCL communicator)
    if (c->d_buf) (void)hipFree(c->d_buf);
    if (c->h_buf) (void)hipHostFree(c->h_buf);
    if (c->d_mc) (void)hipFree(c->d_mc);
    if (c->h_mc) (void)hipHostFree(c->h_mc);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    for (auto e : c->ev_chain) if (e) (void)hipEventDestroy(e);
    if (c->ev_fin) (void)hipEventDestroy(c->ev_fin);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    for (auto e : c->ev_lane) if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

canopy_status comm_alloc(canopy_ctx* ctx, ncclComm_t nc, int world, int rank, canopy_comm** out) {
    canopy_comm* c = new (std::nothrow) canopy_comm();
    if (!c) return CANOPY_ERR_OUT_OF_MEMORY;
    c->ctx = ctx; c->device = ctx->device; c->comm = nc; c->world = world; c->rank = rank;
    if (hipSetDevice(ctx->device) != hipSuccess) {
        (void)hipGetLastError();
        delete c;
        return comm_fail(ctx, CANOPY_ERR_HIP, "hipSetDevice failed");
    }
    const size_t bytes = sizeof(double) * CANOPY_PARTIAL_DOUBLES * (size_t)world;
    hipError_t e = hipMalloc(&c->d_buf, canopy_ctx::kLanes * bytes);
    if (e == hipSuccess) e = hipHostMalloc(&c->h_buf, bytes);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_fin);
    if (e == hipSuccess) e = hipEventCreate(&c->ev_done);
    // Default mode CANOPY_COMM_CHAINED: the collective goes onto the lane that ran the final reduce, behind an event
    // recorded after the communicator's previous collective (canopy_comm_set_mode has the alternatives).
    c->mode = CANOPY_COMM_CHAINED;
    for (int k = 0; k < canopy_ctx::kLanes && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&c->ev_lane[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc(&c->h_ring, bytes * canopy_ctx::kTicketRing);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipEventCreateWithFlags(&c->ev_chain[k], hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc(&c->d_mc, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipHostMalloc(&c->h_mc, 2 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_buf, 0, canopy_ctx::kLanes * bytes, ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        comm_release(c);
        return comm_fail(ctx, CANOPY_ERR_OUT_OF_MEMORY, std::string("comm buffers: ") + hipGetErrorString(e));
    }
    // every later pass on this context fills slot `rank` of its lane's buffer and runs the collective on it
    canopy_status st = canopy_ctx_set_gather_buffer(ctx, c->d_buf, world, rank);
    if (st != CANOPY_OK) { canopy_comm_free(c); return st; }
    for (int k = 0; k < canopy_ctx::kLanes; ++k) ctx->sets[k].d_gather = c->d_buf + (size_t)k * CANOPY_PARTIAL_DOUBLES * (size_t)world;
    for (auto& T : ctx->ring) T.ticket = 0; // (tickets of earlier passes have no combined records)
    ctx->comm = c;
    *out = c;
    return CANOPY_OK;
}

} // namespace

namespace canopy {

// Called by the pass right behind k_finalize: buf = the slot buffer of the pass's lane, lane_stream = the stream its
// final reduce ran on. No host round trip. Consecutive passes of a pipelined sequence run on different lanes, i.e. issue
// the collectives of ONE communicator from different streams; in the default CHAINED mode each collective waits (device
// side) for an event recorded behind the previous one, so the device runs them strictly in issue order -- the same
// order on every rank -- without relying on what RCCL does between streams. The edge sits in the lane's latency-bound
// tail, not in front of the next pass's row kernel.
canopy_status comm_all_reduce_partials(canopy_ctx* ctx, hipStream_t lane_stream, double* buf, int lane, bool timed, int publish_slot) {
    canopy_comm* c = ctx->comm;
    if (!c || c->deferred) return CANOPY_OK;
    hipStream_t st = lane_stream;
    if (c->mode == CANOPY_COMM_OWN_STREAM && c->stream) {
        HIPC_TRY(ctx, hipEventRecord(c->ev_lane[lane], lane_stream));
        HIPC_TRY(ctx, hipStreamWaitEvent(c->stream, c->ev_lane[lane], 0));
        st = c->stream;
    } else if (c->mode == CANOPY_COMM_CHAINED && c->chain_valid && c->chain_stream != st) {
        HIPC_TRY(ctx, hipStreamWaitEvent(st, c->ev_chain[c->chain_next ^ 1], 0)); // behind the previous collective
    }
    // the two events around the collective only for a timed pass (an event record is a packet on the stream and a few
    // microseconds of host time: not on the path of every pass of an 8-way shard)
    if (timed) HIPC_TRY(ctx, hipEventRecord(c->ev_fin, st));
    RCCL_TRY(ctx, rccl()->AllReduce(buf, buf, (size_t)c->world * CANOPY_PARTIAL_DOUBLES, ncclDouble, ncclSum,
                                    static_cast<ncclComm_t>(c->comm), st));
    if (c->mode == CANOPY_COMM_CHAINED) {
        HIPC_TRY(ctx, hipEventRecord(c->ev_chain[c->chain_next], st));
        c->chain_next ^= 1;
        c->chain_stream = st;
        c->chain_valid = true;
    }
    if (publish_slot >= 0) // ticketed pass: the combined records into its slot of the pinned ring
        HIPC_TRY(ctx, canopy::launch_publish(buf, c->world * CANOPY_PARTIAL_DOUBLES,
                                             c->h_ring + (size_t)publish_slot * c->world * CANOPY_PARTIAL_DOUBLES, st));
    if (timed || st != lane_stream) HIPC_TRY(ctx, hipEventRecord(c->ev_done, st));
    c->timed_valid = timed;
    c->issued = true;
    c->last_buf = buf;
    return CANOPY_OK;
}

hipStream_t comm_stream(canopy_ctx* ctx) { // null: collectives ride the lanes
    return (ctx->comm && !ctx->comm->deferred && ctx->comm->mode == CANOPY_COMM_OWN_STREAM) ? ctx->comm->stream : nullptr;
}

// K6: [hits, trials] of every rank summed (one u64 all-reduce of 2 words), same stream.
canopy_status comm_all_reduce_mc(canopy_ctx* ctx, const unsigned long long* d_hits, long long trials) {
    canopy_comm* c = ctx->comm;
    if (!c || c->deferred) return CANOPY_OK; // one process, N devices: the host adds the N hit counts itself
    hipLaunchKernelGGL(k_mc_pack, dim3(1), dim3(1), 0, ctx->stream, c->d_mc, d_hits, (unsigned long long)trials);
    HIPC_TRY(ctx, hipGetLastError());
    hipStream_t st = ctx->stream;
    if (c->mode == CANOPY_COMM_OWN_STREAM && c->stream) {
        HIPC_TRY(ctx, hipEventRecord(c->ev_lane[0], ctx->stream)); // the communicator's own stream, behind the pack
        HIPC_TRY(ctx, hipStreamWaitEvent(c->stream, c->ev_lane[0], 0));
        st = c->stream;
    } else if (c->mode == CANOPY_COMM_CHAINED && c->chain_valid && c->chain_stream != st) {
        HIPC_TRY(ctx, hipStreamWaitEvent(st, c->ev_chain[c->chain_next ^ 1], 0));
    }
    RCCL_TRY(ctx, rccl()->AllReduce(c->d_mc, c->d_mc, 2, ncclUint64, ncclSum, static_cast<ncclComm_t>(c->comm), st));
    if (c->mode == CANOPY_COMM_CHAINED) {
        HIPC_TRY(ctx, hipEventRecord(c->ev_chain[c->chain_next], st));
        c->chain_next ^= 1;
        c->chain_stream = st;
        c->chain_valid = true;
    }
    c->mc_stream = st;
    c->mc_valid = true;
    return CANOPY_OK;
}

} // namespace canopy

extern "C" {

canopy_status canopy_comm_get_unique_id(canopy_comm_id* id) try {
    if (!id) return CANOPY_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(canopy_comm_id) == sizeof(ncclUniqueId), "canopy_comm_id carries an ncclUniqueId");
    RcclApi* api = rccl();
    if (!api->handle) return CANOPY_ERR_RCCL;
    ncclUniqueId u;
    if (api->GetUniqueId(&u) != ncclSuccess) return CANOPY_ERR_RCCL;
    memcpy(id->bytes, &u, sizeof(u));
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_comm_init_rank(canopy_ctx* ctx, const canopy_comm_id* id, int32_t world_size, int32_t rank,
                                    canopy_comm** out) try {
    if (!ctx || !id || !out || world_size < 1 || rank < 0 || rank >= world_size) return CANOPY_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (ctx->comm) return comm_fail(ctx, CANOPY_ERR_STATE, "the context already has a communicator");
    RcclApi* api = rccl();
    if (!api->handle) return comm_fail(ctx, CANOPY_ERR_RCCL, api->error);
    HIPC_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(&u, id->bytes, sizeof(u));
    ncclComm_t nc = nullptr;
    RCCL_TRY(ctx, api->CommInitRank(&nc, world_size, u, rank));
    canopy_status st = comm_alloc(ctx, nc, world_size, rank, out);
    if (st != CANOPY_OK) (void)api->CommDestroy(nc);
    return st;
} CANOPY_CATCH_ALL

canopy_status canopy_comm_init_all(canopy_ctx* const* ctxs, int32_t n, canopy_comm** out) try {
    if (!ctxs || !out || n < 1 || n > 64) return CANOPY_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i) {
        out[i] = nullptr;
        if (!ctxs[i]) return CANOPY_ERR_INVALID_ARGUMENT;
        if (ctxs[i]->comm) return comm_fail(ctxs[i], CANOPY_ERR_STATE, "the context already has a communicator");
        for (int k = 0; k < i; ++k)
            if (ctxs[k]->device == ctxs[i]->device)
                return comm_fail(ctxs[i], CANOPY_ERR_INVALID_ARGUMENT, "canopy_comm_init_all: one context per device");
    }
    RcclApi* api = rccl();
    if (!api->handle) return comm_fail(ctxs[0], CANOPY_ERR_RCCL, api->error);
    int devs[64];
    ncclComm_t ncs[64];
    for (int i = 0; i < n; ++i) { devs[i] = ctxs[i]->device; ncs[i] = nullptr; }
    RCCL_TRY(ctxs[0], api->CommInitAll(ncs, n, devs));
    for (int i = 0; i < n; ++i) {
        canopy_status st = comm_alloc(ctxs[i], ncs[i], n, i, &out[i]);
        if (st != CANOPY_OK) {
            for (int k = 0; k < n; ++k) {
                if (k < i) { canopy_comm_free(out[k]); out[k] = nullptr; }
                else (void)api->CommDestroy(ncs[k]);
            }
            return st;
        }
        out[i]->deferred = true; // the collectives of one pass are issued together (canopy_comm_quantify_all_async)
    }
    return CANOPY_OK;
} CANOPY_CATCH_ALL

void canopy_comm_free(canopy_comm* c) {
    if (!c) return;
    canopy_ctx* ctx = c->ctx;
    if (ctx) { // (null once its context has been freed: canopy_ctx_free takes a forgotten communicator with it)
        (void)hipSetDevice(ctx->device);
        (void)canopy::ctx_join_side(ctx);
        if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
        if (ctx->comm == c) {
            ctx->comm = nullptr;
            (void)canopy_ctx_set_gather_buffer(ctx, nullptr, 0, 0);
            for (auto& S : ctx->sets) S.d_gather = nullptr;
            for (auto& T : ctx->ring) T.ticket = 0; // (their combined records lived in this communicator)
        }
        c->ctx = nullptr;
    } else {
        (void)hipSetDevice(c->device); // (the context went first: its streams were drained when it detached this communicator)
    }
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm && rccl()->handle) (void)rccl()->CommDestroy(static_cast<ncclComm_t>(c->comm));
    comm_release(c);
}

canopy_status canopy_comm_info(const canopy_comm* c, int32_t* world_size, int32_t* rank, int32_t* rccl_version) try {
    if (!c) return CANOPY_ERR_INVALID_ARGUMENT;
    if (world_size) *world_size = c->world;
    if (rank) *rank = c->rank;
    if (rccl_version) {
        int v = 0;
        if (!rccl()->handle || rccl()->GetVersion(&v) != ncclSuccess) v = 0;
        *rccl_version = v;
    }
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_comm_quantify_all_async(canopy_comm* const* comms, const canopy_model* const* models, int32_t n,
                                             const canopy_quant_params* params) try {
    if (!comms || !models || n < 1) return CANOPY_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i)
        if (!comms[i] || !models[i] || !comms[i]->ctx || comms[i]->world != n || comms[i]->rank != i)
            return CANOPY_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < n; ++i) { // row kernels + final reduce on every device's own stream
        const bool was = comms[i]->deferred;
        comms[i]->deferred = true;
        canopy_status st = canopy_quantify_async(comms[i]->ctx, models[i], params);
        comms[i]->deferred = was;
        if (st != CANOPY_OK) return st;
    }
    RcclApi* api = rccl();
    canopy_ctx* c0 = comms[0]->ctx;
    RCCL_TRY(c0, api->GroupStart());
    for (int i = 0; i < n; ++i) {
        canopy_comm* c = comms[i];
        c->issued = true;
        c->last_buf = c->d_buf;
        if (hipSetDevice(c->ctx->device) != hipSuccess) { // each grouped call on its own device
            (void)hipGetLastError();
            (void)api->GroupEnd();
            return comm_fail(c->ctx, CANOPY_ERR_HIP, "hipSetDevice failed inside the grouped all-reduce");
        }
        ncclResult_t r = api->AllReduce(c->d_buf, c->d_buf, (size_t)c->world * CANOPY_PARTIAL_DOUBLES, ncclDouble, ncclSum,
                                        static_cast<ncclComm_t>(c->comm), c->ctx->stream);
        if (r != ncclSuccess) {
            (void)api->GroupEnd();
            return comm_fail(c->ctx, CANOPY_ERR_RCCL, std::string("ncclAllReduce: ") + api->GetErrorString(r));
        }
    }
    RCCL_TRY(c0, api->GroupEnd());
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_comm_fetch(canopy_comm* c, uint32_t flags, canopy_quant_result* result, canopy_partial* parts) try {
    if (!c || !c->ctx || (!result && !parts)) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_ctx* ctx = c->ctx;
    HIPC_TRY(ctx, hipSetDevice(ctx->device));
    if (!c->issued) return comm_fail(ctx, CANOPY_ERR_STATE, "no pass has been enqueued since the communicator was attached");
    const size_t bytes = sizeof(double) * CANOPY_PARTIAL_DOUBLES * (size_t)c->world;
    HIPC_TRY(ctx, canopy::ctx_join_side(ctx)); // a pipelined pass ran its collective on the side stream
    HIPC_TRY(ctx, hipMemcpyAsync(c->h_buf, c->last_buf ? c->last_buf : c->d_buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPC_TRY(ctx, hipStreamSynchronize(ctx->stream));
    static_assert(sizeof(canopy_partial) == sizeof(double) * CANOPY_PARTIAL_DOUBLES, "record layout");
    const canopy_partial* recs = reinterpret_cast<const canopy_partial*>(c->h_buf);
    if (parts) memcpy(parts, recs, bytes);
    if (result) return canopy_combine_partials(recs, c->world, flags, result); // rank order: the same on every rank
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_comm_set_mode(canopy_comm* c, int32_t mode) try {
    if (!c || !c->ctx) return CANOPY_ERR_INVALID_ARGUMENT;
    if (mode != CANOPY_COMM_CHAINED && mode != CANOPY_COMM_UNORDERED && mode != CANOPY_COMM_OWN_STREAM)
        return comm_fail(c->ctx, CANOPY_ERR_INVALID_ARGUMENT, "unknown communicator mode");
    canopy_ctx* ctx = c->ctx;
    HIPC_TRY(ctx, hipSetDevice(ctx->device));
    // nothing in flight while the discipline changes
    HIPC_TRY(ctx, canopy::ctx_join_side(ctx));
    HIPC_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (c->stream) HIPC_TRY(ctx, hipStreamSynchronize(c->stream));
    if (mode == CANOPY_COMM_OWN_STREAM && !c->stream) HIPC_TRY(ctx, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->mode = mode;
    c->chain_valid = false;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

// The combined records of a ticketed pass: its lane published them into pinned host memory right behind the all-reduce;
// this waits for THAT lane only and folds them in rank order.
canopy_status canopy_comm_ticket_fetch(canopy_comm* c, canopy_ticket ticket, uint32_t flags, canopy_quant_result* result,
                                       canopy_partial* parts) try {
    if (!c || !c->ctx || ticket == 0 || (!result && !parts)) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_ctx* ctx = c->ctx;
    if (c->deferred) return comm_fail(ctx, CANOPY_ERR_UNSUPPORTED, "tickets belong to the process-per-GPU shape (canopy_comm_init_rank)");
    const size_t slot = (size_t)(ticket % canopy_ctx::kTicketRing);
    canopy_ctx::TicketSlot& T = ctx->ring[slot];
    if (T.ticket != ticket)
        return comm_fail(ctx, CANOPY_ERR_STATE, "ticket expired or unknown: records stay fetchable until 64 later passes have been enqueued on the context");
    HIPC_TRY(ctx, hipSetDevice(ctx->device));
    HIPC_TRY(ctx, hipEventSynchronize(T.ev));
    canopy_ctx::PassScratch& S = ctx->sets[T.lane];
    if (S.last_seq == ticket) {
        S.pending_lo = 0;
        unsigned long long count = 0;
        memcpy(&count, ctx->h_ring + slot * 2 * CANOPY_PARTIAL_DOUBLES + CANOPY_PARTIAL_DOUBLES, sizeof(count));
        canopy::apply_feedback(ctx, S, count);
    }
    const canopy_partial* recs = reinterpret_cast<const canopy_partial*>(c->h_ring + slot * c->world * CANOPY_PARTIAL_DOUBLES);
    if (parts) memcpy(parts, recs, sizeof(canopy_partial) * (size_t)c->world);
    if (result) return canopy_combine_partials(recs, c->world, flags, result);
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_comm_last_collective_ms(canopy_comm* c, double* ms) try {
    if (!c || !c->ctx || !ms) return CANOPY_ERR_INVALID_ARGUMENT;
    if (!c->issued || c->deferred || !c->timed_valid)
        return comm_fail(c->ctx, CANOPY_ERR_STATE, "no timed collective: the last pass was not timed (canopy_quantify, or a canopy_timing_begin region)");
    HIPC_TRY(c->ctx, hipSetDevice(c->ctx->device));
    HIPC_TRY(c->ctx, hipEventSynchronize(c->ev_done));
    float t = 0.f;
    HIPC_TRY(c->ctx, hipEventElapsedTime(&t, c->ev_fin, c->ev_done));
    *ms = t;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

canopy_status canopy_comm_mc_fetch(canopy_comm* c, canopy_mc_result* r) try {
    if (!c || !c->ctx || !r) return CANOPY_ERR_INVALID_ARGUMENT;
    canopy_ctx* ctx = c->ctx;
    if (!c->mc_valid) return comm_fail(ctx, CANOPY_ERR_STATE, "no Monte-Carlo run since the communicator was attached");
    HIPC_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long host[2];
    hipStream_t st = c->mc_stream ? c->mc_stream : ctx->stream;
    HIPC_TRY(ctx, hipMemcpyAsync(c->h_mc, c->d_mc, sizeof(host), hipMemcpyDeviceToHost, st));
    HIPC_TRY(ctx, hipStreamSynchronize(st));
    host[0] = c->h_mc[0]; host[1] = c->h_mc[1];
    r->hits = (int64_t)host[0];
    r->num_trials = (int64_t)host[1];
    const double n = (double)host[1];
    r->estimate = n > 0 ? (double)host[0] / n : 0.0;
    r->std_error = n > 0 ? sqrt(r->estimate * (1.0 - r->estimate) / n) : 0.0;
    r->elapsed_kernel_ms = -1.0;
    return CANOPY_OK;
} CANOPY_CATCH_ALL

} // extern "C"

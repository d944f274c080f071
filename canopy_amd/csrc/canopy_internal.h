// canopy_internal.h -- host-side state behind the opaque handles of include/canopy.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "canopy.h"
#include "canopy_guard.h"

// Tuning switches and timing ablations of benchmarks/ (environment variables CANOPY_*): compiled in only with
// -DCANOPY_ABLATIONS (CANOPY_ABLATIONS=1 python -m canopy_amd.build). The default build reads NO environment variable:
// some of the ablations give wrong results by design, and a stray variable must never change a quantification.
#ifdef CANOPY_ABLATIONS
#include <stdlib.h>
#define CANOPY_TUNE_ENV(name) getenv(name)
#else
#define CANOPY_TUNE_ENV(name) (static_cast<const char*>(nullptr))
#endif

namespace canopy {
struct BatchState; // kernels_batch.hip
void batch_state_free(BatchState* b);

// One VERSION of a model's probabilities on the device, with the tables derived from it. A model keeps a ring of kPSets
// of them: canopy_model_set_probabilities writes the NEXT set (asynchronously, from a pinned staging buffer) while
// passes still in flight keep reading the sets they were enqueued with -- a new probability vector does not drain the
// lanes of a context. Writes into a set (the upload, a table build) are followed by a record of ev_write on the stream
// they were enqueued on; a pass on any other stream waits for that record once (PassScratch::seen_write).
constexpr int kPSets = 6; // more than the lanes of a context plus the set being written
struct PSet {
    double* d_logp = nullptr;     // [n_events]
    uint8_t* d_q8 = nullptr;      // [n_events rounded up to 16] + 16: quantised -logp of the conservative filter
    uint64_t* d_fx = nullptr;     // [n_events] fixed-point -logp * 2^47 of the FIXED_SUM pass
    double* h_stage = nullptr;    // pinned [n_events]: what the upload copies from
    uint64_t version = 0;         // canopy_model::logp_version of the values in d_logp (0 = never written)
    bool q8_valid = false, fx_valid = false;
    double q8_log_cut = 0.0;      // the log cut-off d_q8 was built for
    hipEvent_t ev_write = nullptr;     // behind the last write into the set (null: an external set, written on the context's stream)
    hipStream_t write_stream = nullptr;// the stream that write went to (passes on it are ordered by the stream itself)
    uint64_t write_id = 0;             // canopy_model::write_counter at that write
    uint64_t reader_ctx = 0;      // id of the context whose passes read the set (0 none, ~0: more than one context)
    uint64_t read_hi = 0;         // highest pass number (canopy_ctx::seq) of that context that read it
    bool external = false;        // buffers owned by somebody else (the envelope of a batch)
};
constexpr uint64_t kManyReaders = ~0ULL;

// host_pool.cpp: persistent worker threads for the host-side log of the probabilities (glibc log IS the canonical logp).
// fn(lo, hi) is called on disjoint ranges covering [0, n); returns when all of them have run.
void host_parallel_for(int64_t n, int64_t grain, const std::function<void(int64_t, int64_t)>& fn);
// out[i] = log(p[i]) (is_log: out[i] = p[i]); returns the index of the first value outside [0, 1] (is_log: above 0 or NaN), -1 if none
int64_t host_log_values(const double* p, double* out, int64_t n, bool is_log);
} // namespace canopy

struct canopy_comm;

struct canopy_ctx {
    uint64_t id = 0; // unique per process (never reused): probability sets remember their readers by it
    uint64_t seq = 0; // pass counter of this context = the ticket of the last enqueued pass
    int device = 0;
    canopy_comm* comm = nullptr; // K5 inside the library (comm_rccl.hip); owned by the caller, detached by canopy_comm_free
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int n_cu = 0;
    int n_blocks_override = 0;
    size_t lds_per_block_max = 0;

    // Lanes. A sum-only FILTER_REFINE / FIXED_SUM pass is a row kernel that streams the whole model (HBM-bound, every CU)
    // followed by a latency-bound tail (exact pass over the listed rows, final reduce, with a communicator the
    // all-reduce). Consecutive such passes rotate through THREE lanes, each a stream of its own with its own scratch,
    // record and all-reduce slot buffer: a pass runs serially on its lane, and the device overlaps the lanes -- the next
    // pass's row kernel takes the CUs the moment this pass's row kernel leaves them, while this pass's tail runs beside
    // it. No event joins the lanes on the hot path (a cross-queue wait or record costs microseconds of queue time even
    // when already satisfied): the host blocks before it reuses a lane until that lane's previous pass has finished.
    // Lane 0 is the context's own stream; passes with per-row outputs and the other kernel variants run there, behind
    // everything the other lanes still hold.
    static constexpr int kLanes = 4; // at most; n_lanes of them rotate (canopy_ctx_set_pipeline: default 4)
    struct PassScratch {
        hipStream_t stream = nullptr;       // lane 0: the context's stream
        double* d_block_partials = nullptr; // [block_partials_cap][8]
        int block_partials_cap = 0;
        uint16_t* d_fr_list = nullptr;      // [n_slabs * slab_rows] per-slab candidate lists (local row ids)
        unsigned* d_fr_cnt = nullptr;       // their lengths
        int64_t fr_list_cap = 0, fr_cnt_cap = 0;
        unsigned long long* d_fr_count = nullptr; // [2]: candidate accumulator of the running pass, total of the finished one
        double* d_partial = nullptr;        // the shard record this lane's final reduce writes (lane 0: ctx->d_partial)
        double* d_gather = nullptr;         // this lane's all-reduce slot buffer when a communicator is attached
        hipEvent_t ev_done = nullptr;       // end of the lane's last pass that recorded one
        bool pending = false;               // work on this lane the context's stream has not been ordered behind yet
        bool used = false;
        uint64_t last_seq = 0;              // pass number of the lane's last pass
        uint64_t pending_lo = 0;            // lowest pass number on this lane not known to have finished (0 = none)
        bool last_has_event = false;        // ev_done was recorded behind the lane's last pass
        // probability sets this lane is already ordered behind (model id, write ids per set)
        uint64_t seen_model = 0;
        uint64_t seen_write[canopy::kPSets] = {};
        // AUTO's feedback (how many rows the conservative pass listed) belongs to the lane that ran the pass
        uint64_t fb_model = 0;              // model id of the lane's last two-step pass (0 = none / consumed)
        int64_t fb_rows = 0;
        double fb_cut_off = -1.0;
        uint64_t fb_version = 0;
        int32_t fb_variant = 0;
    } sets[kLanes];
    // Tickets: the record of a ticketed pass lands in pinned host memory (k_finalize writes it), in slot `pass number mod
    // kTicketRing` of a ring, with an event of its own behind the pass: a ticket stays fetchable until kTicketRing later
    // passes have been enqueued on the context, whatever lanes they ran on (un-pipelined passes share lane 0).
    static constexpr int kTicketRing = 64;
    struct TicketSlot {
        uint64_t ticket = 0;      // pass number whose record the slot holds / will hold; 0 none
        int lane = 0;             // the lane that ran it (AUTO's feedback lives there)
        hipEvent_t ev = nullptr;  // behind the pass (its collective and publish included)
    } ring[kTicketRing];
    double* h_ring = nullptr;     // pinned [kTicketRing][16]: the shard record [0..8), the filter's candidate total as u64 at [8]
    bool pipeline_on = true;      // canopy_ctx_set_pipeline
    int n_lanes = 4;              // lanes consecutive pipelined passes rotate through (2 .. kLanes; round 3, late: four -- 42.7 against 46.3 us per step at the 8-way shard size once the filter had got faster, 0.318 against 0.323 ms for the whole model)
    int pipe_next = 0;            // lane of the next pipelined pass
    int last_set = 0;             // lane of the last pass (its record and candidate count are fetched from there)
    double* d_partial = nullptr;        // the shard record, 8 doubles
    double* h_partial = nullptr;        // pinned mirror
    double* d_gather = nullptr;         // caller-owned all-reduce buffer [world][8], optional
    int gather_world = 0, gather_rank = 0;

    double* d_row_logp = nullptr;       // optional outputs of the last pass
    int64_t row_logp_cap = 0;
    uint64_t* d_keep_mask = nullptr;
    int64_t keep_mask_cap = 0;
    bool last_has_row_logp = false, last_has_mask = false, last_has_importance = false;
    double* d_importance = nullptr;     // [importance_cap] contributions of the last pass that asked for them
    int64_t importance_cap = 0;
    double* ext_importance = nullptr;   // caller-owned buffer instead (cross-GPU all-reduce), ext_importance_n doubles
    int32_t ext_importance_n = 0;

    hipEvent_t ev_last[3] = {nullptr, nullptr, nullptr};
    bool last_timed = false;
    bool want_last_timing = false; // set around the pass of the synchronous canopy_quantify
    // batched quantification ("run only the filter of a FILTER_REFINE pass and leave its candidate lists in lane 0"): what that pass left
    bool filter_only_done = false;
    int filter_only_slab_rows = 0;
    int64_t filter_only_slabs = 0;
    // timing region
    std::vector<hipEvent_t> ev_pool; // 3 per launch
    int timing_max = 0, timing_n = 0;
    bool timing_on = false;

    // FILTER_REFINE: candidates flagged by the last filter pass (device counter, pinned mirror)
    int32_t last_variant = 0;                 // CANOPY_KERNEL_* the last quantify pass ran
    unsigned long long* h_fr_count = nullptr;
    uint64_t fr_known_id = 0;     // last model id with a known candidate fraction
    double fr_known_fraction = -1.0;
    int64_t fr_known_count = 0, fr_known_rows = 0;
    double fr_known_cut_off = -1.0; // the feedback applies to the same (model, p, cut-off) only
    uint64_t fr_known_version = 0;
    int32_t fr_known_variant = 0; // FILTER_REFINE or FIXED_SUM: whose list the count belongs to
    // the same feedback for FIXED_SUM (rows it had to leave to the exact pass), kept apart from the filter's
    uint64_t fx_known_id = 0, fx_known_version = 0;
    double fx_known_fraction = -1.0, fx_known_cut_off = -1.0;
    // K6 Monte-Carlo
    unsigned long long* d_mc_hits = nullptr; // [1]
    unsigned long long* h_mc_hits = nullptr; // pinned
    int64_t mc_last_trials = 0;
    hipEvent_t ev_mc[2] = {nullptr, nullptr};
    // batched quantification (lazily created)
    canopy::BatchState* batch = nullptr;

    uint64_t scratch_gen = 0; // bumped whenever a scratch / output buffer a captured pass may point at is replaced
    bool capturing = false; // inside canopy_graph_capture: no timing events, derived tables always rebuilt

    std::string last_error;
};

struct canopy_comm {
    canopy_ctx* ctx = nullptr; // null once the context has been freed (canopy_ctx_free detaches; the handle stays the caller's to free)
    int device = 0;
    void* comm = nullptr;   // ncclComm_t
    int world = 1, rank = 0;
    double* d_buf = nullptr; // [lanes][world][8] gather slots = the all-reduce buffers, one per lane of the context
    double* h_buf = nullptr; // pinned mirror
    double* h_ring = nullptr; // pinned [canopy_ctx::kTicketRing][world][8]: the combined records of the ticketed passes (slot = pass number mod ring)
    // Ordering of consecutive collectives (canopy_comm_set_mode). CHAINED (default): the collective rides the lane that
    // ran the final reduce, behind an event recorded after the PREVIOUS collective of this communicator -- the device
    // runs the collectives of one communicator strictly in issue order whatever stream they are on. UNORDERED: no such
    // edge (RCCL chains a communicator's launches itself). OWN_STREAM: every collective on `stream`, behind an event of
    // the lane whose final reduce filled the buffer.
    int32_t mode = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev_lane[canopy_ctx::kLanes] = {}; // "this lane's record is in its slot buffer"
    hipEvent_t ev_chain[2] = {nullptr, nullptr}; // behind the last collective, alternating (CHAINED)
    int chain_next = 0;
    hipStream_t chain_stream = nullptr; // the stream the last collective went to
    bool chain_valid = false;
    hipEvent_t ev_fin = nullptr, ev_done = nullptr; // around the last all-reduce (timing)
    bool issued = false;
    bool timed_valid = false; // ev_fin / ev_done bracket the last collective
    double* last_buf = nullptr; // slot buffer of the last issued collective
    unsigned long long* d_mc = nullptr; // [2] {hits, trials} of the Monte-Carlo combine
    unsigned long long* h_mc = nullptr;
    hipStream_t mc_stream = nullptr; // the stream the last Monte-Carlo combine went to
    bool mc_valid = false;
    bool deferred = false;   // one process, N devices: the collectives of a pass are issued together, grouped
};

struct canopy_graph {
    int device = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t model_id = 0;
    const canopy_model* model = nullptr; // must outlive the graph (canopy.h)
    uint64_t scratch_gen = 0;      // canopy_ctx::scratch_gen at capture
    int32_t variant = 0;
    bool has_row_logp = false, has_mask = false, has_importance = false;
};

struct canopy_dag {
    int device = 0;
    int32_t n_events = 0, n_gates = 0, n_levels = 0, top = 0, n_args = 0, vote_planes = 6, n_slots = 0;
    int4* d_desc = nullptr;          // level-ordered gates: {type | K << 8, output node id, arg begin, arg end}
    int32_t* d_args = nullptr;
    int32_t* d_level_ptr = nullptr;  // [n_levels+1] into the level-ordered gate list
    uint64_t* d_thresh = nullptr;    // [n_events] ceil(p * 2^53)
    bool has_p = false;
    // second internal form (kernels_mc.hip, k_mc_reg): helper trees of fan-in 4, the top OR folded into the gates it reads
    bool reg_ok = false, reg_top_folded = false, reg_top_out_flip = false;
    int32_t reg_n_gates = 0, reg_n_levels = 0, reg_n_slots = 0, reg_top = -1, reg_n_args = 0;
    int4* d_reg_desc = nullptr;
    int32_t* d_reg_args = nullptr;
    int32_t* d_reg_level_ptr = nullptr;
    std::vector<int32_t> reg_level_gates; // gates per level (host copy: the launch derives the per-lane schedule from it)
};

struct LsOwned { // the quantised filter's row stream (kernels_ls.hip), owned by the model
    uint32_t* ent = nullptr;   // [n_lines][64] the lanes' bit streams, a range (step, wave) after the other
    uint32_t* bm = nullptr;    // [n_units] item ends of every unit of 32 entry rows
    void* desc = nullptr;      // [n_steps * 16 + 1] ranges: first line, entry rows, first unit
    uint32_t* posinfo = nullptr; // [n_slabs * 8192] list position -> local row id (13 bits) | item length - row length (3 bits; 7 =
                                 // look the bounds up) | row start relative to its slab's first entry (16 bits)
    uint8_t* item_len8 = nullptr;// [n_slabs * 128] length of every item, in list order (position >> 6)
    int64_t n_lines = 0, n_units = 0, n_steps = 0, n_slabs = 0;
    int bits = 0, ss = 0;      // bits per entry; slabs per step
    int64_t stream_bytes = 0, device_bytes = 0;
};

struct FrP24Owned { // device buffers of the 24-bit stream form, owned by the model
    uint32_t* stream = nullptr;
    uint32_t* base = nullptr;
    int32_t* exc = nullptr;
    int64_t n_exception_slabs = 0;
};

// Constness: the C ABI hands quantification a `const canopy_model*` -- the CSR and the probabilities a pass reads are not
// changed by it. What a pass MAY do is build, once, a derived layout or a derived table it finds missing: those members are
// `mutable` and every such build runs under `mu`, so several contexts (and host threads) may quantify one model.
struct canopy_model {
    uint64_t id = 0; // unique per process (never reused): contexts refer to models by id, not by pointer
    int device = 0;
    int64_t n_rows = 0, nnz = 0, row_offset = 0;
    int32_t n_events = 0, max_row_len = 0;
    int64_t* d_row_ptr = nullptr; // [n_rows+1], rebased to 0
    int32_t* d_col_idx = nullptr; // [nnz] (+ 16 B of slack for 16-byte slab loads)
    mutable std::mutex mu;        // lazy builds of derived layouts / derived tables, the bookkeeping of the sets
    mutable canopy::PSet psets[canopy::kPSets];
    int cur = -1;                 // the set canopy_model_set_probabilities wrote last (-1: none yet)
    uint64_t logp_version = 0;    // bumped by every set_probabilities
    mutable uint64_t write_counter = 0;
    mutable int graph_refs = 0;   // captured passes hold the CURRENT set's pointers: while any exists, set_probabilities writes in place
    bool has_p = false;
    mutable int64_t device_bytes = 0;
    // derived wave-slab stream layout (kernels_ws.hip), built by canopy_model_prepare
    mutable uint32_t* d_ws_stream = nullptr;
    mutable int64_t* d_ws_off = nullptr;
    mutable int64_t ws_total_chunks = 0, n_ws = 0;
    mutable int ws_state = 0; // 0 not built, 1 ready, -1 not representable (fall back to ROWS)
    // derived band stream (kernels_bs.hip): the exact all-rows kernel with workgroup-wide slabs
    mutable uint32_t* d_bs_stream = nullptr;
    mutable int64_t* d_bs_off = nullptr;   // [n_bs_slabs + 1] first superstep of every slab (allotted, not all used)
    mutable int32_t* d_bs_used = nullptr;  // [n_bs_slabs] supersteps the slab uses
    mutable int64_t bs_total_ss = 0, bs_used_ss = 0, n_bs_slabs = 0;
    mutable int bs_state = 0; // 0 not built, 1 ready, -1 not representable, -2 representable only with > 30 % padding (explicit requests)
    // derived value-scatter layout (kernels_vs.hip)
    mutable uint32_t* d_vs_stream = nullptr;
    mutable int64_t* d_vs_off = nullptr;
    mutable int64_t* d_vs_slab_row = nullptr;
    mutable int64_t vs_total_chunks = 0, n_slabs = 0;
    mutable int vs_state = 0;
    // derived filter/refine slab stream (kernels_fr.hip)
    mutable uint32_t* d_fr_stream = nullptr;
    mutable int64_t* d_fr_off = nullptr;
    mutable int64_t fr_total_chunks = 0, n_fr_slabs = 0;
    mutable FrP24Owned fr24; // 24-bit form of the same chunks (quantised filter), or nulls
    mutable uint16_t* d_off16 = nullptr; // [n_rows + 1] row start minus the start of the row's 8192-row block (null: some block holds > 65535 entries)
    mutable int fr_state = 0;
    mutable int fr_chunks = 0; // 1: the chunk stream (d_fr_stream and / or fr24) exists; 0 with fr_state 1: the row stream serves the
                               // filter, the chunks are built by the first pass that reads them (ensure_fr_chunks)
    mutable LsOwned ls;       // the row stream of the quantised filter (E <= 112 K), built with the slab stream
    mutable int ls_state = 0; // 0 not built, 1 ready, -1 the model has none
};

namespace canopy {

struct RowKernelArgs {
    const int64_t* row_ptr;
    const int32_t* col_idx;
    const double* logp;
    int64_t n_rows;
    int64_t row_offset;
    int32_t n_events;
    double log_cut;
    int32_t limit_order;
    double* block_partials; // [grid][8]
    double* row_logp_out;   // nullable
    uint64_t* keep_mask_out;// nullable
    const uint16_t* off16 = nullptr; // nullable: row starts relative to their 8192-row block (exact pass of the two-step variants)
};

// kernels_rows.hip
int rows_kernel_default_grid(int n_cu, int variant, int64_t n_rows);
size_t rows_lds_table_limit_events();
hipError_t launch_rows_kernel(int variant, const RowKernelArgs& a, int grid, hipStream_t s);
// fr_count: null, or the FILTER_REFINE candidate counter pair -- [0] (accumulated by the filter) is handed
// over to [1] (read by the host) and cleared for the next pass
// host_rec: null, or the lane's pinned record (ticketed passes): the shard record and the candidate total land there too
hipError_t launch_finalize(const double* block_partials, int n_blocks, double* out, double* gather,
                           int world, int rank, unsigned long long* fr_count, double* host_rec, hipStream_t s);
// copies n doubles from a device buffer into pinned host memory (the combined records behind the all-reduce of a ticketed pass)
hipError_t launch_publish(const double* src, int n, double* host_dst, hipStream_t s);

// kernels_build.hip
hipError_t launch_importance(const uint64_t* keep_mask, int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx,
                             const double* logp, double* contrib, hipStream_t s);
hipError_t launch_validate_csr(const int64_t* row_ptr, const int32_t* col_idx, int64_t n_rows,
                               int64_t nnz, int32_t n_events, int64_t* d_status /*[2]: first bad row, max len*/,
                               hipStream_t s);
hipError_t launch_rebase_row_ptr(int64_t* row_ptr, int64_t n_plus_1, int64_t base, hipStream_t s);
hipError_t scan_inclusive_i64(int64_t* data, int64_t n, hipStream_t s);
hipError_t synth_row_ptr(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t avg_nnz, int64_t** d_row_ptr_out,
                         hipStream_t s);
hipError_t synth_build_csr(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t n_events,
                           int32_t avg_nnz, int64_t** d_row_ptr_out, int32_t** d_col_idx_out,
                           int64_t* nnz_out, hipStream_t s);


// quantify_pass.cpp -- the one pass of the path: policy, scratch, launches
struct PassOptions {
    bool filter_only = false;       // batch path: run only the conservative filter of FILTER_REFINE, leave its lists in lane 0
    uint64_t* ticket_out = nullptr; // non-null: the pass is ticketed (its record lands in pinned host memory)
};
canopy_status enqueue_pass(canopy_ctx* ctx, const canopy_model* m, PSet* P, const canopy_quant_params* params, const PassOptions& opt);
// makes lane stream `lane` (scratch S) ordered behind the last write into set `slot` of m
hipError_t order_behind_set(canopy_ctx::PassScratch& S, hipStream_t lane, const canopy_model* m, int slot);
// blocks the host until no pass of ctx (or anybody's, if the readers are unknown) still reads P
hipError_t wait_set_readers(canopy_ctx* ctx, PSet& P);
// consumes the AUTO feedback of lane S given the candidate total its last two-step pass published
void apply_feedback(canopy_ctx* ctx, canopy_ctx::PassScratch& S, unsigned long long count);
// lazy derived layouts (under m->mu)
canopy_status build_fr(canopy_ctx* ctx, const canopy_model* m);
canopy_status ensure_fr32(canopy_ctx* ctx, const canopy_model* m);
canopy_status ensure_fr_chunks(canopy_ctx* ctx, const canopy_model* m);
canopy_status build_vs(canopy_ctx* ctx, const canopy_model* m);
canopy_status build_ws(canopy_ctx* ctx, const canopy_model* m);
canopy_status build_bs(canopy_ctx* ctx, const canopy_model* m, bool explicit_request);
int auto_variant(const canopy_model* m);
int64_t fr_min_rows(int base_variant, bool sequence);
canopy_status fail(canopy_ctx* ctx, canopy_status st, const std::string& msg);
uint64_t next_object_id();

// comm_rccl.hip
// the single collective of a pass, enqueued on the stream its final reduce ran on; publish: also copy the combined
// records into the lane's pinned buffer (ticketed pass)
canopy_status comm_all_reduce_partials(canopy_ctx* ctx, hipStream_t lane_stream, double* buf, int lane, bool timed, int publish_slot);
hipStream_t comm_stream(canopy_ctx* ctx);
// orders the context's stream behind everything enqueued on the other lanes (device-side waits, no host blocking)
hipError_t ctx_join_side(canopy_ctx* ctx);
canopy_status comm_all_reduce_mc(canopy_ctx* ctx, const unsigned long long* d_hits, long long trials);

// kernels_ws.hip
int ws_rows_per_wave();
int ws_max_events();
int ws_default_grid(int n_cu, int64_t n_ws);
hipError_t ws_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int32_t n_events,
                    int32_t max_row_len, uint32_t** d_stream_out, int64_t** d_ws_off_out,
                    int64_t* total_chunks_out, int64_t* n_ws_out, bool* ok, hipStream_t s);
hipError_t launch_ws_kernel(const RowKernelArgs& ra, const uint32_t* stream, const int64_t* ws_off,
                            int64_t total_chunks, int64_t n_ws, int need_len, int grid, int* n_records,
                            hipStream_t s);

// kernels_bs.hip
int bs_max_events();
int bs_slab_rows();
int64_t bs_ss_bytes();
int bs_default_grid(int n_cu, int64_t n_slabs);
hipError_t bs_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int64_t nnz, int32_t n_events,
                    int32_t max_row_len, bool accept_padding, uint32_t** d_stream_out, int64_t** d_off_out, int32_t** d_used_out,
                    int64_t* total_ss_out, int64_t* used_ss_out, int64_t* n_slabs_out, bool* ok, hipStream_t s);
hipError_t launch_bs_kernel(const RowKernelArgs& ra, const uint32_t* stream, const int64_t* ss_off, const int32_t* ss_used, int64_t total_ss,
                            int64_t n_slabs, int need_len, int grid, int* n_records, hipStream_t s);

// kernels_vs.hip
int vs_max_events();
int vs_max_row_len();
int vs_default_grid(int n_cu, int64_t n_slabs);
hipError_t vs_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int64_t nnz, int32_t n_events,
                    int32_t max_row_len, uint32_t** d_stream_out, int64_t** d_vs_off_out, int64_t** d_slab_row_out,
                    int64_t* total_chunks_out, int64_t* n_slabs_out, bool* ok, hipStream_t s);
hipError_t launch_vs_kernel(const RowKernelArgs& ra, const uint32_t* stream, const int64_t* vs_off,
                            const int64_t* slab_row, int64_t total_chunks, int64_t n_slabs, int grid, hipStream_t s);


// kernels_fr.hip
int fr_max_events();
int fr_refine_grid(int n_cu);
int fr_parts(int32_t n_events); // column parts of the slab stream (1 up to 112 K events): offsets are per (slab, part)
hipError_t fr_offsets(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int32_t n_events, int64_t** d_fr_off_out,
                      int64_t* total_chunks_out, int64_t* n_slabs_out, bool* ok, hipStream_t s);
hipError_t fr_build_sorted32(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int32_t n_events,
                             const int64_t* d_fr_off, int64_t total_chunks, int64_t n_slabs, uint32_t** d_stream_out,
                             hipStream_t s);
int fr_slab_rows(int32_t n_events);
// 2-byte row offsets for the exact pass (null when a block of 8192 rows holds more than 65535 entries); waits for the stream
hipError_t fr_build_off16(const int64_t* d_row_ptr, int64_t n_rows, uint16_t** d_off16_out, hipStream_t s);
hipError_t launch_fr_filter(const uint32_t* stream, const int64_t* fr_off, int64_t total_chunks, const double* logp,
                            int64_t n_rows, int32_t n_events, int64_t n_slabs, double cand_cut, uint16_t* cand_list,
                            unsigned* cand_cnt, unsigned long long* cand_count, int n_cu, hipStream_t s);
bool fr_filter_is_quantised(int32_t n_events);
// the 24-bit form of the slab stream (kernels_fr.hip, "P24"): what the quantised filter reads when the model allows it
struct FrP24 {
    const uint32_t* stream = nullptr; // 768-byte chunks at the 32-bit stream's chunk indices
    const uint32_t* base = nullptr;   // first column of every chunk
    const int32_t* exc = nullptr;     // [n_slabs] 1 = exception slab (all rows go to the exact pass)
};
bool fr_p24_applies(int32_t n_events);
hipError_t fr_build_pack24(const uint32_t* d_stream32, const int64_t* d_fr_off, int64_t n_slabs, int64_t total_chunks, int32_t n_events,
                           FrP24Owned* p, hipStream_t s);
// the row stream of the quantised filter (kernels_ls.hip): lists of POSITIONS, translated by posinfo in the exact pass
struct LsView {
    const uint32_t* ent = nullptr;
    const uint32_t* bm = nullptr;
    const void* desc = nullptr;
    const uint32_t* posinfo = nullptr;
    const uint8_t* item_len8 = nullptr;
    int64_t n_steps = 0, n_slabs = 0;
    int bits = 0, ss = 0;
    bool alone = false; // the pass is not one of a pipelined sequence: the filter reads its stream with non-temporal loads
};
bool ls_applies(int32_t n_events);
void ls_free(LsOwned* L);
hipError_t ls_build(const int64_t* d_row_ptr, const int32_t* d_col_idx, int64_t n_rows, int64_t nnz, int32_t n_events, int n_cu, LsOwned* out,
                    bool* ok, hipStream_t s);
hipError_t launch_ls_filter(const LsView& L, const uint8_t* q8, int32_t n_events, bool nib, bool alone, int threshold, uint16_t* cand_list,
                            unsigned* cand_cnt, unsigned long long* cand_count, int n_cu, hipStream_t s);
hipError_t launch_fr_q8_table(const double* logp, int32_t n_events, double log_cut, uint8_t* q8_buf, hipStream_t s);
bool fr_q8_overlapped(const FrP24* p24, int32_t n_events, int32_t max_row_len);
void fr_q8_overlapped_units(int* unit_rows, int* units_per_slab);
hipError_t launch_fr_filter_q8(const uint32_t* stream, const FrP24* p24, bool overlapped, int32_t max_row_len, const int64_t* fr_off,
                               int64_t total_chunks, const double* logp, uint8_t* q8_buf, int64_t n_rows, int32_t n_events, int64_t n_slabs,
                               double log_cut, uint16_t* cand_list, unsigned* cand_cnt, unsigned long long* cand_count,
                               int n_cu, bool table_valid, hipStream_t s, const LsView* ls = nullptr);
// flagged: the lists come from launch_fx_sum (flag bits 15 / 14 per entry); records go to block_partials[part_offset + block]
hipError_t launch_fr_refine(const RowKernelArgs& a, const uint16_t* cand_list, const unsigned* cand_cnt, int slab_rows,
                            int64_t n_slabs, uint64_t* keep_mask, int grid, bool flagged, int part_offset, hipStream_t s,
                            const uint32_t* posinfo = nullptr, const uint8_t* item_len8 = nullptr);
int fx_max_row_len();
int fx_sum_grid(int n_cu, int32_t n_events, int64_t n_slabs);
void fx_list_units(int32_t n_events, int* unit_rows, int* units_per_slab);
hipError_t launch_fx_table(const double* logp, int32_t n_events, uint64_t* fx, hipStream_t s);
hipError_t launch_fx_sum(const uint32_t* stream, const int64_t* fr_off, int64_t total_chunks, const uint64_t* fx,
                         const int64_t* row_ptr, int64_t n_rows, int32_t n_events, int64_t n_slabs, double log_cut,
                         int32_t limit_order, int32_t max_row_len, double* block_partials, uint16_t* cand_list,
                         unsigned* cand_cnt, unsigned long long* cand_count, uint64_t* keep_mask, int grid, hipStream_t s);

} // namespace canopy

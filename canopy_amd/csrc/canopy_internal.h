// canopy_internal.h -- host-side state behind the opaque handles of include/canopy.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

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
} // namespace canopy

struct canopy_comm;

struct canopy_ctx {
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
    static constexpr int kLanes = 3;
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
        hipEvent_t ev_done = nullptr;       // end of the lane's last pass
        bool pending = false;               // work on this lane the context's stream has not been ordered behind yet
        bool used = false;
    } sets[kLanes];
    bool pipeline_on = true;      // canopy_ctx_set_pipeline / CANOPY_NO_PIPELINE
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
    // batched quantification: "run only the filter of a FILTER_REFINE pass and leave its candidate lists in set 0"
    bool filter_only_request = false, filter_only_done = false;
    int filter_only_slab_rows = 0;
    int64_t filter_only_slabs = 0;
    // timing region
    std::vector<hipEvent_t> ev_pool; // 3 per launch
    int timing_max = 0, timing_n = 0;
    bool timing_on = false;

    // FILTER_REFINE: candidates flagged by the last filter pass (device counter, pinned mirror)
    int32_t last_variant = 0;                 // CANOPY_KERNEL_* the last quantify pass ran
    unsigned long long* h_fr_count = nullptr;
    uint64_t fr_pending_id = 0;   // model id whose filter count is still on the device (0 = none)
    int64_t fr_pending_rows = 0;
    uint64_t fr_known_id = 0;     // last model id with a known candidate fraction
    double fr_known_fraction = -1.0;
    int64_t fr_known_count = 0, fr_known_rows = 0;
    double fr_known_cut_off = -1.0, fr_pending_cut_off = -1.0; // the feedback applies to the same (model, p, cut-off) only
    uint64_t fr_known_version = 0, fr_pending_version = 0;
    int32_t fr_pending_variant = 0, fr_known_variant = 0; // FILTER_REFINE or FIXED_SUM: whose list the count belongs to
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
    canopy_ctx* ctx = nullptr;
    void* comm = nullptr;   // ncclComm_t
    int world = 1, rank = 0;
    double* d_buf = nullptr; // [lanes][world][8] gather slots = the all-reduce buffers, one per lane of the context
    double* h_buf = nullptr; // pinned mirror
    // CANOPY_COMM_OWN_STREAM=1 only: every collective is enqueued on `stream`, behind an event of the lane whose final
    // reduce filled the buffer (default: null, the collective rides the lane; RCCL orders calls from different streams)
    hipStream_t stream = nullptr;
    hipEvent_t ev_lane[canopy_ctx::kLanes] = {nullptr, nullptr, nullptr}; // "this lane's record is in its slot buffer"
    hipEvent_t ev_fin = nullptr, ev_done = nullptr; // around the last all-reduce (timing)
    bool issued = false;
    bool timed_valid = false; // ev_fin / ev_done bracket the last collective
    double* last_buf = nullptr; // slot buffer of the last issued collective
    unsigned long long* d_mc = nullptr; // [2] {hits, trials} of the Monte-Carlo combine
    unsigned long long* h_mc = nullptr;
    bool mc_valid = false;
    bool deferred = false;   // one process, N devices: the collectives of a pass are issued together, grouped
};

struct canopy_graph {
    int device = 0;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t model_id = 0;
    canopy_model* model = nullptr; // must outlive the graph (canopy.h)
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
};

struct FrP24Owned { // device buffers of the 24-bit stream form, owned by the model
    uint32_t* stream = nullptr;
    uint32_t* base = nullptr;
    int32_t* exc = nullptr;
    int64_t n_exception_slabs = 0;
};

struct canopy_model {
    uint64_t id = 0; // unique per process (never reused): contexts refer to models by id, not by pointer
    int device = 0;
    int64_t n_rows = 0, nnz = 0, row_offset = 0;
    int32_t n_events = 0, max_row_len = 0;
    int64_t* d_row_ptr = nullptr; // [n_rows+1], rebased to 0
    int32_t* d_col_idx = nullptr; // [nnz] (+ 16 B of slack for 16-byte slab loads)
    double* d_logp = nullptr;     // [n_events]
    uint8_t* d_q8 = nullptr;      // [n_events rounded up to 16] quantised -logp of the FILTER_REFINE filter pass
    uint64_t logp_version = 0;    // bumped whenever d_logp's contents (or the pointer) change
    uint64_t* d_fx = nullptr;     // [n_events] fixed-point -logp * 2^47 of the FIXED_SUM pass
    uint64_t fx_version = 0;      // logp_version d_fx was built from (0 = none)
    uint64_t q8_version = 0;      // logp_version the table in d_q8 was built from (0 = none)
    double q8_log_cut = 0.0;      // ... and the log cut-off it was built for
    bool has_p = false;
    int64_t device_bytes = 0;
    // derived wave-slab stream layout (kernels_ws.hip), built by canopy_model_prepare
    uint32_t* d_ws_stream = nullptr;
    int64_t* d_ws_off = nullptr;
    int64_t ws_total_chunks = 0, n_ws = 0;
    int ws_state = 0; // 0 not built, 1 ready, -1 not representable (fall back to ROWS)
    // derived value-scatter layout (kernels_vs.hip)
    uint32_t* d_vs_stream = nullptr;
    int64_t* d_vs_off = nullptr;
    int64_t* d_vs_slab_row = nullptr;
    int64_t vs_total_chunks = 0, n_slabs = 0;
    int vs_state = 0;
    // derived filter/refine slab stream (kernels_fr.hip)
    uint32_t* d_fr_stream = nullptr;
    int64_t* d_fr_off = nullptr;
    int64_t fr_total_chunks = 0, n_fr_slabs = 0;
    FrP24Owned fr24; // 24-bit form of the same chunks (quantised filter), or nulls
    int fr_state = 0;
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
};

// kernels_rows.hip
int rows_kernel_default_grid(int n_cu, int variant, int64_t n_rows);
size_t rows_lds_table_limit_events();
hipError_t launch_rows_kernel(int variant, const RowKernelArgs& a, int grid, hipStream_t s);
// fr_count: null, or the FILTER_REFINE candidate counter pair -- [0] (accumulated by the filter) is handed
// over to [1] (read by the host) and cleared for the next pass
hipError_t launch_finalize(const double* block_partials, int n_blocks, double* out, double* gather,
                           int world, int rank, unsigned long long* fr_count, hipStream_t s);

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


// comm_rccl.hip
// the single collective of a pass, enqueued on the stream its final reduce ran on
canopy_status comm_all_reduce_partials(canopy_ctx* ctx, hipStream_t lane_stream, double* buf, int lane, bool timed);
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
bool fr_q8_overlapped(const FrP24* p24, int32_t n_events, int32_t max_row_len);
void fr_q8_overlapped_units(int* unit_rows, int* units_per_slab);
hipError_t launch_fr_filter_q8(const uint32_t* stream, const FrP24* p24, bool overlapped, int32_t max_row_len, const int64_t* fr_off,
                               int64_t total_chunks, const double* logp, uint8_t* q8_buf, int64_t n_rows, int32_t n_events, int64_t n_slabs,
                               double log_cut, uint16_t* cand_list, unsigned* cand_cnt, unsigned long long* cand_count,
                               int n_cu, bool table_valid, hipStream_t s);
// flagged: the lists come from launch_fx_sum (flag bits 15 / 14 per entry); records go to block_partials[part_offset + block]
hipError_t launch_fr_refine(const RowKernelArgs& a, const uint16_t* cand_list, const unsigned* cand_cnt, int slab_rows,
                            int64_t n_slabs, uint64_t* keep_mask, int grid, bool flagged, int part_offset, hipStream_t s);
int fx_max_row_len();
int fx_sum_grid(int n_cu, int32_t n_events, int64_t n_slabs);
void fx_list_units(int32_t n_events, int* unit_rows, int* units_per_slab);
hipError_t launch_fx_table(const double* logp, int32_t n_events, uint64_t* fx, hipStream_t s);
hipError_t launch_fx_sum(const uint32_t* stream, const int64_t* fr_off, int64_t total_chunks, const uint64_t* fx,
                         const int64_t* row_ptr, int64_t n_rows, int32_t n_events, int64_t n_slabs, double log_cut,
                         int32_t limit_order, int32_t max_row_len, double* block_partials, uint16_t* cand_list,
                         unsigned* cand_cnt, unsigned long long* cand_count, uint64_t* keep_mask, int grid, hipStream_t s);

} // namespace canopy

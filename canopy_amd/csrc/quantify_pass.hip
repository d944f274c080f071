// quantify_pass.hip -- ONE quantification pass over a model: what runs (policy), what it needs (scratch), how it is
// launched (lanes, tickets, the collective). Host code only; the kernels live in kernels_*.hip.
//
// The reference has no quantifier (src/canopy.cpp:1-31 is a licence comment); the knobs a pass honours are the ones of
// mef::openpsa::Settings (src/mef/openpsa/settings.h:19,22,314-332: approximation, cut_off, limit_order), the input is
// BasicEvent::p() (src/mef/openpsa/event/event.h:93-96) as a probability SET of the model (canopy::PSet).
//
// Structure (round 3: the former 300-line canopy_quantify_async, split):
//   plan_pass        AUTO policy + validation: which kernel variant, which outputs, pipelined or not  -> PassPlan
//   ensure_layouts   the derived layout the variant reads (lazy, one-off, under the model's mutex)
//   ensure_outputs   per-row / mask / importance buffers of the context
//   ensure_lane      scratch of the lane the pass runs on
//   ensure_tables    the set's derived tables (8-bit filter table, fixed-point table), built on the pass's lane
//   launch_pass      the launches: row kernel(s), final reduce, collective, record publication, events
#include <math.h>
#include <string.h>

#include <atomic>
#include <new>

#include "canopy_device.h"
#include "canopy_internal.h"

namespace canopy {

uint64_t next_object_id() { // contexts and models may be created from different host threads
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

// ---- policy: what AUTO resolves to ------------------------------------------------------------------------------

// What AUTO resolves to for this model (before knowing whether the WS build succeeds).
int auto_variant(const canopy_model* m) {
    // measured on MI355X (avg 6 nnz/row, 1e8 rows): the LDS-table row kernel wins up to E = 2048 (two
    // workgroups per CU still fit); beyond that the table costs occupancy and the wave-slab stream wins
    // (round 3: at scale the band stream is ahead even there -- 1e8 rows, s_i written, E = 1e3 avg 4 / E = 2048 avg 6: ROWS_LDS 0.76 /
    // 0.99 ms, BAND_STREAM 0.66 / 0.83, benchmarks/bs_small_e.py -- but it works in rounds of one 16384-row slab per CU (~40 us
    // each): below ~2e7 rows the two are level and ROWS_LDS has no floor)
    if (m->n_events <= 2048) return m->n_rows >= (int64_t)20000000 ? CANOPY_KERNEL_BAND_STREAM : CANOPY_KERNEL_ROWS_LDS;
    // round 2, E = 1e5, every row kept: ROWS 64 us / 113 us at 2e6 / 4e6 rows, WAVE_STREAM 72 / 75 (a ~50 us floor: few
    // wave-slabs, one wave each): the stream kernel from 2.5e6 rows
    // round 3: the band stream (workgroup-wide slabs) where its 17-bit columns reach: 0.93 against 1.19 ms per pass with s_i
    // written at config 3 (ensure_layouts falls back to the wave-slab stream for a model the band layout does not fit); a
    // ~44 us floor (one slab = one workgroup): level with ROWS at ~1.4e6 rows for E = 1e5, ~2e6 for E = 1e4
    // (benchmarks/bs_small.py: 2e6 rows, s_i written: ROWS 64 / 49 us, WAVE_STREAM 57 / 52, BAND_STREAM 49 / 48)
    if (m->n_events <= bs_max_events() && m->n_rows >= (int64_t)2000000) return CANOPY_KERNEL_BAND_STREAM;
    if (m->n_events <= ws_max_events() && m->n_rows >= (int64_t)2500000) return CANOPY_KERNEL_WAVE_STREAM;
    return CANOPY_KERNEL_ROWS;
}

// Sizes from which the two slab passes replace the row kernel AUTO would run (round 2, pipelined and serial passes at
// E = 1e3 and 1e5: filter/refine is level with the row kernels at ~4e5 .. 7e5 rows serially and ahead from 1e5 rows in a
// pipelined sequence; the fixed-point pass is level with ROWS_LDS at ~3e6 rows, with ROWS / WAVE_STREAM at ~1e6)
// (end of round 3, with the row stream and one slab per step on small models, benchmarks/crossover.py: a pass alone -- three short
// kernels, 37-39 us whatever the size -- is level with ROWS at ~3.5e5 rows (E = 1e5) and with ROWS_LDS at ~5.5e5 (E = 1e3); in a
// pipelined sequence of passes it is ahead from the smallest sizes measured: 13-17 us per step against 21-22 at 5e4 .. 2e5 rows)
int64_t fr_min_rows(int base_variant, bool sequence) {
    if (sequence) return (int64_t)100000;
    return base_variant == CANOPY_KERNEL_ROWS_LDS ? (int64_t)600000 : (int64_t)400000;
}
static int64_t fx_min_rows(int base_variant) { return base_variant == CANOPY_KERNEL_ROWS_LDS ? (int64_t)4000000 : (int64_t)1000000; }

// ---- lazy derived layouts (one-off per model, independent of p; every build runs under m->mu and ends with a host
// ---- wait, so that ANY stream of any context may read the layout afterwards) -------------------------------------

// the column-sorted chunks of a model whose offsets are known (m->mu held): the 32-bit stream, its 24-bit form where the model
// allows one (the 32-bit form then goes again: ensure_fr32 brings it back for the passes that read it)
static canopy_status build_fr_chunks_locked(canopy_ctx* ctx, const canopy_model* m) {
    auto drop = [&](hipError_t e, const char* what) { // a failed build leaves no half-made stream behind (the offsets stay)
        (void)hipGetLastError();
        if (m->d_fr_stream) (void)hipFree(m->d_fr_stream);
        m->d_fr_stream = nullptr;
        return fail(ctx, e == hipErrorOutOfMemory ? CANOPY_ERR_OUT_OF_MEMORY : CANOPY_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    hipError_t e = fr_build_sorted32(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, m->d_fr_off, m->fr_total_chunks,
                                     m->n_fr_slabs, &m->d_fr_stream, ctx->stream);
    if (e != hipSuccess) return drop(e, "fr_build_sorted32");
    e = fr_build_pack24(m->d_fr_stream, m->d_fr_off, m->n_fr_slabs, m->fr_total_chunks, m->n_events, &m->fr24, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        if (m->fr24.stream) (void)hipFree(m->fr24.stream);
        if (m->fr24.base) (void)hipFree(m->fr24.base);
        if (m->fr24.exc) (void)hipFree(m->fr24.exc);
        m->fr24 = FrP24Owned();
        return drop(e, "fr_build_pack24");
    }
    m->device_bytes += (m->fr_total_chunks + 1) * 1024;
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
    m->fr_chunks = 1;
    return CANOPY_OK;
}

// What the two slab passes read. The chunk offsets, the exact pass's 2-byte row offsets and -- up to 260 096 events -- the
// quantised filter's row stream (kernels_ls.hip) are built here; the column-sorted CHUNKS themselves (14 ms and 1.8 GB at
// config 3) only for a model without a row stream: a sum-only FILTER_REFINE pass never reads them, and the passes that do
// (FIXED_SUM, the batch path, which walks the lists itself) build them when they first run (ensure_fr_chunks).
canopy_status build_fr(canopy_ctx* ctx, const canopy_model* m) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->fr_state != 0) return CANOPY_OK;
    bool ok = false;
    HIP_TRY(ctx, fr_offsets(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, &m->d_fr_off, &m->fr_total_chunks, &m->n_fr_slabs, &ok, ctx->stream));
    if (!ok) { m->fr_state = -1; return CANOPY_OK; }
    m->device_bytes += (m->n_fr_slabs * fr_parts(m->n_events) + 1) * 8;
    hipError_t e = fr_build_off16(m->d_row_ptr, m->n_rows, &m->d_off16, ctx->stream); // 2 B per row: what the exact pass reads instead of row_ptr
    if (e != hipSuccess) { (void)hipGetLastError(); m->d_off16 = nullptr; }  // (optional layout: the pass runs without it)
    if (m->d_off16) m->device_bytes += (m->n_rows + 1) * 2;
    if (m->ls_state == 0) {
        bool ls_ok = false;
        e = ls_build(m->d_row_ptr, m->d_col_idx, m->n_rows, m->nnz, m->n_events, ctx->n_cu, &m->ls, &ls_ok, ctx->stream);
        if (e != hipSuccess) { (void)hipGetLastError(); ls_ok = false; }
        m->ls_state = ls_ok ? 1 : -1;
        if (ls_ok) m->device_bytes += m->ls.device_bytes;
    }
    static const bool eager = CANOPY_TUNE_ENV("CANOPY_FR_EAGER_CHUNKS") != nullptr; // tuning only
    if (m->ls_state != 1 || eager) {
        canopy_status st = build_fr_chunks_locked(ctx, m);
        if (st != CANOPY_OK) { // nothing half-made stays behind
            if (m->d_fr_off) (void)hipFree(m->d_fr_off);
            m->d_fr_off = nullptr; m->fr_total_chunks = 0; m->n_fr_slabs = 0;
            if (m->d_off16) { (void)hipFree(m->d_off16); m->d_off16 = nullptr; }
            ls_free(&m->ls);
            m->ls_state = 0;
            return st;
        }
    }
    m->fr_state = 1;
    return CANOPY_OK;
}

// the chunks of a model whose build_fr left them to the first pass that reads them
canopy_status ensure_fr_chunks(canopy_ctx* ctx, const canopy_model* m) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->fr_state != 1 || m->fr_chunks == 1) return CANOPY_OK;
    return build_fr_chunks_locked(ctx, m);
}

// the 32-bit slab stream of a model whose build_fr gave it up for the 24-bit form
canopy_status ensure_fr32(canopy_ctx* ctx, const canopy_model* m) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->fr_state != 1 || m->fr_chunks != 1 || m->d_fr_stream != nullptr) return CANOPY_OK;
    HIP_TRY(ctx, fr_build_sorted32(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, m->d_fr_off, m->fr_total_chunks,
                                   m->n_fr_slabs, &m->d_fr_stream, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    m->device_bytes += (m->fr_total_chunks + 1) * 1024;
    return CANOPY_OK;
}

canopy_status build_vs(canopy_ctx* ctx, const canopy_model* m) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->vs_state != 0) return CANOPY_OK;
    bool ok = false;
    HIP_TRY(ctx, vs_build(m->d_row_ptr, m->d_col_idx, m->n_rows, m->nnz, m->n_events, m->max_row_len, &m->d_vs_stream,
                          &m->d_vs_off, &m->d_vs_slab_row, &m->vs_total_chunks, &m->n_slabs, &ok, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    m->vs_state = ok ? 1 : -1;
    if (ok) m->device_bytes += (m->vs_total_chunks + 1) * 1024 + (m->n_slabs + 1) * 16;
    return CANOPY_OK;
}

canopy_status build_ws(canopy_ctx* ctx, const canopy_model* m) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->ws_state != 0) return CANOPY_OK;
    if (m->n_events > ws_max_events()) { m->ws_state = -1; return CANOPY_OK; }
    bool ok = false;
    HIP_TRY(ctx, ws_build(m->d_row_ptr, m->d_col_idx, m->n_rows, m->n_events, m->max_row_len, &m->d_ws_stream,
                          &m->d_ws_off, &m->ws_total_chunks, &m->n_ws, &ok, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    m->ws_state = ok ? 1 : -1;
    if (ok) m->device_bytes += (m->ws_total_chunks + 1) * 1024 + (m->n_ws + 1) * 8;
    return CANOPY_OK;
}

canopy_status build_bs(canopy_ctx* ctx, const canopy_model* m, bool explicit_request) {
    std::lock_guard<std::mutex> lk(m->mu);
    if (m->bs_state == 1 || m->bs_state == -1 || (m->bs_state == -2 && !explicit_request)) return CANOPY_OK;
    bool ok = false;
    HIP_TRY(ctx, bs_build(m->d_row_ptr, m->d_col_idx, m->n_rows, m->nnz, m->n_events, m->max_row_len, explicit_request, &m->d_bs_stream,
                          &m->d_bs_off, &m->d_bs_used, &m->bs_total_ss, &m->bs_used_ss, &m->n_bs_slabs, &ok, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    m->bs_state = ok ? 1 : (explicit_request ? -1 : -2);
    if (ok) m->device_bytes += (m->bs_total_ss + 1) * bs_ss_bytes() + (m->n_bs_slabs + 1) * 8 + m->n_bs_slabs * 4;
    return CANOPY_OK;
}

// ---- probability sets: ordering between the stream that wrote a set and the lanes that read it -------------------

hipError_t order_behind_set(canopy_ctx::PassScratch& S, hipStream_t lane, const canopy_model* m, int slot) {
    std::lock_guard<std::mutex> lk(m->mu);
    const PSet& P = m->psets[slot];
    if (S.seen_model != m->id) {
        S.seen_model = m->id;
        for (auto& w : S.seen_write) w = 0;
    }
    if (S.seen_write[slot] == P.write_id) return hipSuccess;
    if (P.write_stream != lane && P.ev_write != nullptr && P.write_id != 0) {
        hipError_t e = hipStreamWaitEvent(lane, P.ev_write, 0);
        if (e != hipSuccess) return e;
    }
    S.seen_write[slot] = P.write_id;
    return hipSuccess;
}

hipError_t wait_set_readers(canopy_ctx* ctx, PSet& P) {
    hipError_t e = hipSuccess;
    if (P.reader_ctx != 0 && P.reader_ctx != ctx->id) {
        // passes of another context (or of several) read this set: nothing finer than the device is known from here
        e = hipDeviceSynchronize();
    } else if (P.read_hi != 0) {
        for (int k = 0; k < canopy_ctx::kLanes && e == hipSuccess; ++k) {
            canopy_ctx::PassScratch& S = ctx->sets[k];
            if (S.pending_lo == 0 || S.pending_lo > P.read_hi) continue; // nothing that old is still in flight on this lane
            e = S.last_has_event ? hipEventSynchronize(S.ev_done) : hipStreamSynchronize(S.stream);
            if (e == hipSuccess) S.pending_lo = 0;
        }
    }
    if (e == hipSuccess) { P.reader_ctx = 0; P.read_hi = 0; }
    return e;
}

void apply_feedback(canopy_ctx* ctx, canopy_ctx::PassScratch& S, unsigned long long count) {
    if (S.fb_model == 0) return;
    const double frac = S.fb_rows > 0 ? (double)count / (double)S.fb_rows : 0.0;
    if (S.fb_variant == CANOPY_KERNEL_FIXED_SUM) {
        ctx->fx_known_id = S.fb_model;
        ctx->fx_known_cut_off = S.fb_cut_off;
        ctx->fx_known_version = S.fb_version;
        ctx->fx_known_fraction = frac;
    } else {
        ctx->fr_known_id = S.fb_model;
        ctx->fr_known_cut_off = S.fb_cut_off;
        ctx->fr_known_version = S.fb_version;
        ctx->fr_known_fraction = frac;
    }
    ctx->fr_known_variant = S.fb_variant;
    ctx->fr_known_count = (int64_t)count;
    ctx->fr_known_rows = S.fb_rows;
    S.fb_model = 0;
}

namespace {

// Orders the context's stream behind the other lanes' work (device-side wait, no host blocking).
canopy_status join_side(canopy_ctx* ctx) {
    HIP_TRY(ctx, ctx_join_side(ctx));
    return CANOPY_OK;
}

struct PassPlan {
    int variant = 0;
    bool want_rows = false, want_mask = false, want_imp = false, keep_mask_flag = false;
    bool two_step = false, pipelined = false, filter_only = false, ticketed = false;
    int grid = 0, grid_fx = 0;
    int64_t n_words = 0;
    double log_cut = 0.0;
};

// Validation + AUTO. Builds the layout AUTO's choice depends on (the slab stream), nothing else.
canopy_status plan_pass(canopy_ctx* ctx, const canopy_model* m, const PSet* P, const canopy_quant_params* params,
                        const PassOptions& opt, PassPlan* out) {
    PassPlan pl;
    pl.want_rows = (params->flags & CANOPY_QUANT_ROW_LOGP) != 0;
    pl.want_imp = (params->flags & CANOPY_QUANT_IMPORTANCE) != 0;
    pl.keep_mask_flag = (params->flags & CANOPY_QUANT_KEEP_MASK) != 0;
    pl.want_mask = pl.keep_mask_flag || pl.want_imp; // importance walks the exact keep mask
    pl.filter_only = opt.filter_only;
    pl.ticketed = opt.ticket_out != nullptr;
    pl.n_words = (m->n_rows + 63) / 64;
    pl.log_cut = params->cut_off > 0.0 ? log(params->cut_off) : -INFINITY; // host glibc log, as the oracle
    int variant = params->kernel == CANOPY_KERNEL_AUTO ? auto_variant(m) : params->kernel;
    // AUTO: with a positive cut-off and no per-row output, filter/refine replaces the row kernels once the
    // model is large enough to amortise its launches (fr_min_rows), unless its filter listed too many of
    // the rows last time (fr_leave_at).
    // Where filter/refine stops paying depends on what takes over: against the fixed-point pass (0.67 ms per 1e8 rows
    // whatever the survivors) the crossover is at ~15 % of the rows listed (round 3, config 3, pipelined steps, cut-offs
    // 1e-13 .. 1e-16: 0.597 ms at 10.3 % listed, 0.621 at 12.3 %, 0.650 at 14.4 %, 0.713 at 16.3 %; round 2: 10 % -- the exact
    // pass has since got cheaper per listed row), against the exact CSR-order stream kernel (1.1-1.5 ms) at ~20 %.
    const bool fx_can_take_over = m->n_events >= 1 && m->n_events <= fr_max_events() && m->n_rows >= fx_min_rows(variant) &&
                                  m->max_row_len <= fx_max_row_len() &&
                                  !(ctx->fx_known_id == m->id && ctx->fx_known_fraction > 0.2 &&
                                    ctx->fx_known_cut_off == params->cut_off && ctx->fx_known_version == P->version);
    const double fr_leave_at = fx_can_take_over ? 0.15 : 0.2;
    // (an asynchronous sum-only pass on a pipelining context is taken to be one of a sequence: the synchronous canopy_quantify is not)
    const bool sequence = ctx->pipeline_on && !ctx->want_last_timing && !pl.want_mask && !pl.want_imp && !ctx->capturing;
    if (params->kernel == CANOPY_KERNEL_AUTO && params->cut_off > 0.0 && !pl.want_rows &&
        m->n_events >= 1 && m->n_events <= fr_max_events() && m->n_rows >= fr_min_rows(variant, sequence) &&
        !(ctx->fr_known_id == m->id && ctx->fr_known_fraction > fr_leave_at && ctx->fr_known_cut_off == params->cut_off &&
          ctx->fr_known_version == P->version)) {
        canopy_status pst = build_fr(ctx, m);
        if (pst != CANOPY_OK) return pst;
        if (m->fr_state == 1) variant = CANOPY_KERNEL_FILTER_REFINE;
    }
    // AUTO, sum-only pass that filter/refine does not serve (cut_off = 0, or a cut-off most rows survive): the
    // order-independent fixed-point pass instead of the exact CSR-order stream kernel, unless IT had to hand > 20 % of
    // the rows to the exact pass last time (huge |s|, clamped entries)
    // (also for E <= 2048 at scale, where the row kernel with the table in LDS would run: 1e8 rows, E = 1e3, cut_off 0:
    // 1.01 ms against 0.54 with the fixed-point table in LDS; 1e7 rows: 0.117 against 0.088)
    if (params->kernel == CANOPY_KERNEL_AUTO && !pl.want_rows && fx_can_take_over && variant != CANOPY_KERNEL_FILTER_REFINE) {
        canopy_status pst = build_fr(ctx, m);
        if (pst != CANOPY_OK) return pst;
        if (m->fr_state == 1) variant = CANOPY_KERNEL_FIXED_SUM;
    }
    if (variant == CANOPY_KERNEL_FILTER_REFINE || variant == CANOPY_KERNEL_FIXED_SUM) {
        if (variant == CANOPY_KERNEL_FILTER_REFINE && !(params->cut_off > 0.0))
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "FILTER_REFINE needs cut_off > 0");
        if (pl.want_rows)
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "FILTER_REFINE / FIXED_SUM have no per-row output (bit-exact s_i needs the CSR order)");
        if (variant == CANOPY_KERNEL_FIXED_SUM && m->max_row_len > fx_max_row_len())
            return fail(ctx, CANOPY_ERR_UNSUPPORTED, "FIXED_SUM needs rows of at most 63 entries");
    } else if (variant != CANOPY_KERNEL_ROWS && variant != CANOPY_KERNEL_ROWS_LDS && variant != CANOPY_KERNEL_WAVE_STREAM &&
               variant != CANOPY_KERNEL_VALUE_SCATTER && variant != CANOPY_KERNEL_BAND_STREAM) {
        return fail(ctx, CANOPY_ERR_UNSUPPORTED, "unknown kernel variant");
    }
    if (variant == CANOPY_KERNEL_ROWS_LDS && (size_t)m->n_events > rows_lds_table_limit_events())
        return fail(ctx, CANOPY_ERR_UNSUPPORTED, "logp table does not fit the LDS budget of ROWS_LDS");
    if (pl.want_imp && ctx->ext_importance && ctx->ext_importance_n != m->n_events)
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "importance buffer was set for another number of basic events");
    pl.variant = variant;
    *out = pl;
    return CANOPY_OK;
}

// The derived layout the chosen variant reads. May change the variant (AUTO falls back to ROWS when a model cannot be
// laid out as a stream).
canopy_status ensure_layouts(canopy_ctx* ctx, const canopy_model* m, const canopy_quant_params* params, PassPlan& pl) {
    if (pl.variant == CANOPY_KERNEL_FILTER_REFINE || pl.variant == CANOPY_KERNEL_FIXED_SUM) {
        canopy_status pst = build_fr(ctx, m);
        if (pst != CANOPY_OK) return pst;
        if (m->fr_state != 1) return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a filter/refine stream");
        // the row stream serves the quantised filter of a pass that does not hand its lists out: no chunk is read
        const bool ls_serves = pl.variant == CANOPY_KERNEL_FILTER_REFINE && m->ls_state == 1 && !pl.filter_only &&
                               fr_filter_is_quantised(m->n_events) && fr_slab_rows(m->n_events) == 8192;
        if (!ls_serves) {
            if (m->fr_chunks != 1) {
                if (ctx->capturing) return fail(ctx, CANOPY_ERR_STATE, "the slab stream must exist before a capture");
                pst = ensure_fr_chunks(ctx, m);
                if (pst != CANOPY_OK) return pst;
            }
            // the 32-bit form of the stream, if this pass reads it and the model gave it up for the 24-bit one
            const bool reads_p24 = pl.variant == CANOPY_KERNEL_FILTER_REFINE && fr_filter_is_quantised(m->n_events) && m->fr24.stream != nullptr;
            if (!reads_p24 && m->d_fr_stream == nullptr) {
                if (ctx->capturing) return fail(ctx, CANOPY_ERR_STATE, "the slab stream must exist before a capture");
                pst = ensure_fr32(ctx, m);
                if (pst != CANOPY_OK) return pst;
            }
        }
    }
    if (pl.variant == CANOPY_KERNEL_VALUE_SCATTER) {
        canopy_status pst = build_vs(ctx, m);
        if (pst != CANOPY_OK) return pst;
        if (m->vs_state != 1) {
            if (params->kernel == CANOPY_KERNEL_VALUE_SCATTER)
                return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a value-scatter stream");
            pl.variant = CANOPY_KERNEL_ROWS;
        }
    }
    if (pl.variant == CANOPY_KERNEL_BAND_STREAM) {
        canopy_status pst = build_bs(ctx, m, params->kernel == CANOPY_KERNEL_BAND_STREAM);
        if (pst != CANOPY_OK) return pst;
        if (m->bs_state != 1) {
            if (params->kernel == CANOPY_KERNEL_BAND_STREAM)
                return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a band stream");
            pl.variant = m->n_events <= 2048 ? CANOPY_KERNEL_ROWS_LDS // AUTO: what it ran before the band stream existed
                                             : (m->n_events <= ws_max_events() ? CANOPY_KERNEL_WAVE_STREAM : CANOPY_KERNEL_ROWS);
        }
    }
    if (pl.variant == CANOPY_KERNEL_WAVE_STREAM) {
        canopy_status pst = build_ws(ctx, m);
        if (pst != CANOPY_OK) return pst;
        if (m->ws_state != 1) {
            if (params->kernel == CANOPY_KERNEL_WAVE_STREAM)
                return fail(ctx, CANOPY_ERR_UNSUPPORTED, "model cannot be laid out as a wave-slab stream");
            pl.variant = CANOPY_KERNEL_ROWS;
        }
    }
    return CANOPY_OK;
}

// Output buffers of the context (per-row log-probabilities, keep mask, importance contributions).
canopy_status ensure_outputs(canopy_ctx* ctx, const canopy_model* m, const PassPlan& pl) {
    if (pl.want_imp && !ctx->ext_importance && ctx->importance_cap < m->n_events) {
        if (ctx->d_importance) (void)hipFree(ctx->d_importance);
        ctx->d_importance = nullptr; ctx->importance_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_importance, sizeof(double) * (size_t)(m->n_events > 0 ? m->n_events : 1)));
        ctx->importance_cap = m->n_events; ++ctx->scratch_gen;
    }
    if (pl.want_rows && ctx->row_logp_cap < m->n_rows) {
        if (ctx->d_row_logp) (void)hipFree(ctx->d_row_logp);
        ctx->d_row_logp = nullptr; ctx->row_logp_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_row_logp, sizeof(double) * (size_t)(m->n_rows > 0 ? m->n_rows : 1)));
        ctx->row_logp_cap = m->n_rows; ++ctx->scratch_gen;
    }
    if (pl.want_mask && ctx->keep_mask_cap < pl.n_words) {
        if (ctx->d_keep_mask) (void)hipFree(ctx->d_keep_mask);
        ctx->d_keep_mask = nullptr; ctx->keep_mask_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->d_keep_mask, sizeof(uint64_t) * (size_t)(pl.n_words > 0 ? pl.n_words : 1)));
        ctx->keep_mask_cap = pl.n_words; ++ctx->scratch_gen;
    }
    return CANOPY_OK;
}

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

// Scratch of the lane: per-workgroup records; for the two-step passes the candidate lists, their lengths, the counter pair.
canopy_status ensure_lane(canopy_ctx* ctx, canopy_ctx::PassScratch& S, int set, const canopy_model* m, const PassPlan& pl) {
    canopy_status st = ensure_block_partials(ctx, S, pl.variant == CANOPY_KERNEL_WAVE_STREAM ? pl.grid * 4 : pl.grid + pl.grid_fx);
    if (st != CANOPY_OK) return st;
    if (!pl.two_step) return CANOPY_OK;
    if (!S.d_fr_count) {
        // {accumulator of the running pass, total of the last finished pass}
        HIP_TRY(ctx, hipMalloc(&S.d_fr_count, 4 * sizeof(unsigned long long)));
        if (!ctx->h_fr_count) HIP_TRY(ctx, hipHostMalloc(&ctx->h_fr_count, sizeof(unsigned long long)));
        // cleared once: a pass's final reduce leaves the accumulator at zero for the next pass on this set
        HIP_TRY(ctx, hipMemsetAsync(S.d_fr_count, 0, 4 * sizeof(unsigned long long), S.stream));
    }
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
    return CANOPY_OK;
}

// The set's derived table this pass reads, built on the pass's own lane when it is missing -- build, event record and
// bookkeeping in one critical section of the model's mutex, so that a second context taking the same set either builds
// the table itself or finds a write to wait for. A set that was just written has no reader yet, so the build needs no
// drain; only a cut-off change on a set other lanes may still be reading does (rare: join every lane, wait on the host).
// A captured pass rebuilds its table inside the graph at every replay (p may change between replays): *q8_in_graph.
canopy_status ensure_tables(canopy_ctx* ctx, const canopy_model* m, PSet* P, int slot, canopy_ctx::PassScratch& S, const PassPlan& pl,
                            bool* q8_in_graph) {
    *q8_in_graph = false;
    hipStream_t lane = S.stream;
    const size_t ev = (size_t)(m->n_events > 0 ? m->n_events : 1);
    const bool need_fx = pl.variant == CANOPY_KERNEL_FIXED_SUM;
    const bool need_q8 = pl.variant == CANOPY_KERNEL_FILTER_REFINE && fr_filter_is_quantised(m->n_events);
    if (!need_fx && !need_q8) return CANOPY_OK;
    std::lock_guard<std::mutex> lk(m->mu);
    bool wrote = false;
    if (need_fx) {
        if (!P->d_fx) HIP_TRY(ctx, hipMalloc(&P->d_fx, sizeof(uint64_t) * ev));
        if (ctx->capturing || !P->fx_valid) {
            HIP_TRY(ctx, launch_fx_table(P->d_logp, m->n_events, P->d_fx, lane));
            P->fx_valid = true;
            wrote = true;
        }
    } else {
        if (!P->d_q8) HIP_TRY(ctx, hipMalloc(&P->d_q8, ((ev + 15) / 16) * 16 + 16));
        if (ctx->capturing) { // the replayed pass rewrites the table with ITS cut-off: the eager cache key is void
            *q8_in_graph = true;
            P->q8_valid = false;
            return CANOPY_OK;
        }
        if (!(P->q8_valid && P->q8_log_cut == pl.log_cut)) { // the 8-bit table depends on (logp, cut-off) only
            if (P->q8_valid) { // another cut-off on a set that may have readers on other lanes
                HIP_TRY(ctx, ctx_join_side(ctx));
                HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
                if (P->reader_ctx != 0 && P->reader_ctx != ctx->id) HIP_TRY(ctx, hipDeviceSynchronize());
            }
            HIP_TRY(ctx, launch_fr_q8_table(P->d_logp, m->n_events, pl.log_cut, P->d_q8, lane));
            P->q8_valid = true;
            P->q8_log_cut = pl.log_cut;
            wrote = true;
        }
    }
    if (wrote && !P->external && P->ev_write != nullptr && !ctx->capturing) { // later readers on other streams wait for this record
        HIP_TRY(ctx, hipEventRecord(P->ev_write, lane));
        P->write_stream = lane;
        P->write_id = ++m->write_counter;
        if (slot >= 0) S.seen_write[slot] = P->write_id;
    }
    return CANOPY_OK;
}

canopy_status launch_pass(canopy_ctx* ctx, const canopy_model* m, PSet* P, int slot, const canopy_quant_params* params,
                          PassPlan& pl, const PassOptions& opt) {
    // Pipelined pass: a sum-only FILTER_REFINE / FIXED_SUM pass takes the next lane (a stream with scratch, record and slot
    // buffer of its own) and runs there from start to end; the device overlaps the lanes. Everything else runs on lane 0,
    // the context's stream, ordered behind whatever the other lanes still hold.
    const int set = pl.pipelined ? ctx->pipe_next : 0;
    canopy_ctx::PassScratch& S = ctx->sets[set];
    canopy_status st = CANOPY_OK;
    if (!pl.pipelined) {
        st = join_side(ctx);
        if (st != CANOPY_OK) return st;
    } else if (S.used) {
        // the lane's previous pass (its collective included) must have finished: back-pressure on the HOST (it stays up to
        // two passes ahead of the device), not a wait in a device queue
        HIP_TRY(ctx, hipEventSynchronize(S.ev_done));
        S.pending = false;
        S.pending_lo = 0;
    }
    hipStream_t lane = S.stream; // every launch of this pass
    double* const d_gather = S.d_gather ? S.d_gather : ctx->d_gather;
    st = ensure_lane(ctx, S, set, m, pl);
    if (st != CANOPY_OK) return st;
    if (slot >= 0 && !ctx->capturing) HIP_TRY(ctx, order_behind_set(S, lane, m, slot)); // the upload of this set
    bool q8_in_graph = false;
    st = ensure_tables(ctx, m, P, slot, S, pl, &q8_in_graph);
    if (st != CANOPY_OK) return st;
    if (slot >= 0 && !ctx->capturing) HIP_TRY(ctx, order_behind_set(S, lane, m, slot)); // a table another context built meanwhile

    RowKernelArgs a;
    a.row_ptr = m->d_row_ptr; a.col_idx = m->d_col_idx; a.logp = P->d_logp;
    a.n_rows = m->n_rows; a.row_offset = m->row_offset; a.n_events = m->n_events;
    a.log_cut = pl.log_cut;
    a.limit_order = params->limit_order;
    a.block_partials = S.d_block_partials;
    a.row_logp_out = pl.want_rows ? ctx->d_row_logp : nullptr;
    a.keep_mask_out = pl.want_mask ? ctx->d_keep_mask : nullptr;
    a.off16 = pl.two_step ? m->d_off16 : nullptr;

    // HIP events around the pass only when somebody will read them (a timing region, or the synchronous canopy_quantify):
    // every record is a marker packet between two kernels, ~5 us of queue time at the 1.25e7-row shard of an 8-GPU run
    hipEvent_t* ev = ctx->ev_last;
    bool timed = ctx->want_last_timing && !ctx->capturing;
    if (ctx->timing_on && !ctx->capturing && ctx->timing_n < ctx->timing_max) {
        ev = &ctx->ev_pool[(size_t)ctx->timing_n * 3];
        ++ctx->timing_n;
        timed = true;
    }
    const size_t mask_bytes = sizeof(uint64_t) * (size_t)(pl.n_words > 0 ? pl.n_words : 1);
    int n_records = pl.grid;
    int grid = pl.grid;
    if (pl.variant == CANOPY_KERNEL_VALUE_SCATTER && pl.want_mask) // slabs start at arbitrary rows: the kernel ORs partial words into a cleared mask
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_keep_mask, 0, mask_bytes, ctx->stream));
    if (timed) HIP_TRY(ctx, hipEventRecord(ev[0], lane));
    S.fb_model = 0; // (whatever feedback an earlier pass on this lane left unfetched is void now)
    if (pl.two_step) {
        ctx->last_set = set;
        const int slab_rows = fr_slab_rows(m->n_events);
        if (pl.want_mask && pl.variant == CANOPY_KERNEL_FILTER_REFINE && !pl.filter_only) // the refine pass ORs the exact keep bits into a cleared mask
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_keep_mask, 0, mask_bytes, ctx->stream));
        bool q8_overlapped = false; // the quantised filter ran with the end step folded into the next iterations
        bool use_ls = false;        // ... over the row stream: its lists hold positions
        if (pl.variant == CANOPY_KERNEL_FIXED_SUM) {
            // pass A writes records [0, grid_fx) and every word of the keep mask; the exact pass adds records [grid_fx, ..)
            HIP_TRY(ctx, launch_fx_sum(m->d_fr_stream, m->d_fr_off, m->fr_total_chunks, P->d_fx, m->d_row_ptr, m->n_rows, m->n_events,
                                       m->n_fr_slabs, a.log_cut, params->limit_order, m->max_row_len, S.d_block_partials,
                                       S.d_fr_list, S.d_fr_cnt, S.d_fr_count, pl.want_mask ? ctx->d_keep_mask : nullptr,
                                       pl.grid_fx, lane));
        } else if (fr_filter_is_quantised(m->n_events)) {
            FrP24 p24;
            p24.stream = m->fr24.stream; p24.base = m->fr24.base; p24.exc = m->fr24.exc;
            q8_overlapped = fr_q8_overlapped(&p24, m->n_events, m->max_row_len);
            // the row stream when the model has one (its lists hold positions: the batch path, which reads the lists itself,
            // stays on the chunks)
            LsView lsv;
            use_ls = m->ls_state == 1 && !pl.filter_only && fr_slab_rows(m->n_events) == 8192; // (= ensure_layouts' ls_serves)
            if (use_ls) {
                lsv.ent = m->ls.ent; lsv.bm = m->ls.bm; lsv.desc = m->ls.desc; lsv.posinfo = m->ls.posinfo; lsv.item_len8 = m->ls.item_len8;
                lsv.n_steps = m->ls.n_steps; lsv.n_slabs = m->ls.n_slabs; lsv.bits = m->ls.bits; lsv.ss = m->ls.ss;
                lsv.alone = !pl.pipelined || ctx->want_last_timing; // (the synchronous canopy_quantify is a pass alone too)
            }
            HIP_TRY(ctx, launch_fr_filter_q8(m->d_fr_stream, &p24, q8_overlapped, m->max_row_len, m->d_fr_off, m->fr_total_chunks, P->d_logp, P->d_q8, m->n_rows,
                                             m->n_events, m->n_fr_slabs, a.log_cut, S.d_fr_list, S.d_fr_cnt, S.d_fr_count,
                                             ctx->n_cu, !q8_in_graph, lane, use_ls ? &lsv : nullptr));
        } else {
            // delta covers the rounding difference between the unordered pass-A sum and the CSR-order sum
            const double cand_cut = a.log_cut - (1e-9 + 1e-11 * (double)m->max_row_len);
            HIP_TRY(ctx, launch_fr_filter(m->d_fr_stream, m->d_fr_off, m->fr_total_chunks, P->d_logp, m->n_rows,
                                          m->n_events, m->n_fr_slabs, cand_cut, S.d_fr_list, S.d_fr_cnt, S.d_fr_count, ctx->n_cu,
                                          lane));
        }
        int unit_rows = slab_rows, units_per_slab = 1; // list units the exact pass walks
        if (pl.variant == CANOPY_KERNEL_FIXED_SUM) fx_list_units(m->n_events, &unit_rows, &units_per_slab);
        else if (use_ls) { unit_rows = 8192; units_per_slab = 1; }
        else if (q8_overlapped) fr_q8_overlapped_units(&unit_rows, &units_per_slab);
        if (pl.filter_only && pl.variant == CANOPY_KERNEL_FILTER_REFINE) {
            // the batch path lists, marks and compacts the candidates itself (kernels_batch.hip): nothing else to run
            ctx->filter_only_done = true;
            ctx->filter_only_slab_rows = unit_rows;
            ctx->filter_only_slabs = m->n_fr_slabs * units_per_slab;
            ctx->last_variant = pl.variant;
            ctx->last_set = set;
            return CANOPY_OK;
        }
        S.fb_model = m->id; S.fb_rows = m->n_rows; S.fb_cut_off = params->cut_off; S.fb_version = P->version; S.fb_variant = pl.variant;
        const int64_t n_units = m->n_fr_slabs * units_per_slab;
        const int64_t g_need = (2 * n_units + 3) / 4 > 0 ? (2 * n_units + 3) / 4 : 1; // two waves per unit, four per workgroup
        if ((int64_t)grid > g_need) grid = (int)g_need; // a small shard does not pay for 2048 workgroups' records
        HIP_TRY(ctx, launch_fr_refine(a, S.d_fr_list, S.d_fr_cnt, unit_rows, n_units,
                                      pl.want_mask ? ctx->d_keep_mask : nullptr, grid, pl.variant == CANOPY_KERNEL_FIXED_SUM, pl.grid_fx,
                                      lane, use_ls ? m->ls.posinfo : nullptr, use_ls ? m->ls.item_len8 : nullptr));
        n_records = grid + pl.grid_fx;
    } else if (pl.variant == CANOPY_KERNEL_VALUE_SCATTER) {
        HIP_TRY(ctx, launch_vs_kernel(a, m->d_vs_stream, m->d_vs_off, m->d_vs_slab_row, m->vs_total_chunks, m->n_slabs,
                                      grid, ctx->stream));
    } else if (pl.variant == CANOPY_KERNEL_BAND_STREAM) {
        const int need_len = (params->limit_order != 0 && params->limit_order < m->max_row_len) ? 1 : 0;
        HIP_TRY(ctx, launch_bs_kernel(a, m->d_bs_stream, m->d_bs_off, m->d_bs_used, m->bs_total_ss, m->n_bs_slabs, need_len, grid,
                                      &n_records, ctx->stream));
    } else if (pl.variant == CANOPY_KERNEL_WAVE_STREAM) {
        const int need_len = (params->limit_order != 0 && params->limit_order < m->max_row_len) ? 1 : 0;
        HIP_TRY(ctx, launch_ws_kernel(a, m->d_ws_stream, m->d_ws_off, m->ws_total_chunks, m->n_ws, need_len, grid,
                                      &n_records, ctx->stream));
    } else {
        HIP_TRY(ctx, launch_rows_kernel(pl.variant, a, grid, ctx->stream));
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ev[1], lane));
    const bool publish = pl.ticketed && !ctx->capturing;
    const int slot_t = (int)((ctx->seq + 1) % canopy_ctx::kTicketRing); // this pass's slot of the ticket ring
    HIP_TRY(ctx, launch_finalize(S.d_block_partials, n_records, S.d_partial, d_gather, ctx->gather_world, ctx->gather_rank,
                                 pl.two_step ? S.d_fr_count : nullptr,
                                 publish ? ctx->h_ring + (size_t)slot_t * 2 * CANOPY_PARTIAL_DOUBLES : nullptr, lane));
    ctx->last_variant = pl.variant;
    ctx->last_set = set;
    hipStream_t last = lane; // the stream the pass ends on
    if (ctx->comm) { // K5: the one collective of the path, issued here (not by the host language), behind the final reduce
        canopy_status cst = comm_all_reduce_partials(ctx, lane, d_gather, set, timed, publish ? slot_t : -1);
        if (cst != CANOPY_OK) return cst;
        if (comm_stream(ctx)) last = comm_stream(ctx);
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ev[2], last));
    const uint64_t seq = ++ctx->seq;
    S.last_seq = seq;
    if (S.pending_lo == 0) S.pending_lo = seq;
    S.last_has_event = false;
    if ((pl.pipelined || last != lane || publish) && !ctx->capturing) { // someone has to wait for this pass later: the lane's next pass, a fetch, a join
        HIP_TRY(ctx, hipEventRecord(S.ev_done, last));
        S.used = true;
        S.pending = set != 0 || last != lane;
        S.last_has_event = true;
    }
    canopy_ctx::TicketSlot& T = ctx->ring[slot_t]; // (the pass 64 back loses its ticket here, fetched or not)
    T.ticket = publish ? seq : 0;
    T.lane = set;
    if (publish) HIP_TRY(ctx, hipEventRecord(T.ev, last));
    if (opt.ticket_out) *opt.ticket_out = publish ? seq : 0;
    if (!ctx->capturing) { // who reads the set, and up to which pass: what the next writer of the set waits for
        if (P->reader_ctx == 0 || P->reader_ctx == ctx->id) { P->reader_ctx = ctx->id; P->read_hi = seq; }
        else P->reader_ctx = kManyReaders;
    }
    if (pl.pipelined) ctx->pipe_next = (set + 1) % ctx->n_lanes;
    if (pl.want_imp) { // after the timed pass: one walk over the kept rows
        double* dst = ctx->ext_importance ? ctx->ext_importance : ctx->d_importance;
        HIP_TRY(ctx, hipMemsetAsync(dst, 0, sizeof(double) * (size_t)(m->n_events > 0 ? m->n_events : 1), ctx->stream));
        HIP_TRY(ctx, launch_importance(ctx->d_keep_mask, m->n_rows, m->d_row_ptr, m->d_col_idx, P->d_logp, dst, ctx->stream));
    }
    ctx->last_has_importance = pl.want_imp;
    if (ev == ctx->ev_last) ctx->last_timed = timed;
    ctx->last_has_row_logp = pl.want_rows;
    ctx->last_has_mask = pl.keep_mask_flag;
    return CANOPY_OK;
}

} // namespace

canopy_status enqueue_pass(canopy_ctx* ctx, const canopy_model* m, PSet* P, const canopy_quant_params* params, const PassOptions& opt) {
    if (opt.ticket_out) *opt.ticket_out = 0;
    if (params->limit_order < 0 || !(params->cut_off >= 0.0))
        return fail(ctx, CANOPY_ERR_INVALID_ARGUMENT, "limit_order < 0 or cut_off < 0");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    PassPlan pl;
    canopy_status st = plan_pass(ctx, m, P, params, opt, &pl);
    if (st != CANOPY_OK) return st;
    st = ensure_layouts(ctx, m, params, pl);
    if (st != CANOPY_OK) return st;
    st = ensure_outputs(ctx, m, pl);
    if (st != CANOPY_OK) return st;

    if (pl.variant == CANOPY_KERNEL_WAVE_STREAM)
        pl.grid = ctx->n_blocks_override > 0 ? ctx->n_blocks_override : ws_default_grid(ctx->n_cu, m->n_ws);
    else if (pl.variant == CANOPY_KERNEL_BAND_STREAM)
        pl.grid = ctx->n_blocks_override > 0 ? ctx->n_blocks_override : bs_default_grid(ctx->n_cu, m->n_bs_slabs);
    else if (pl.variant == CANOPY_KERNEL_VALUE_SCATTER)
        pl.grid = ctx->n_blocks_override > 0 ? ctx->n_blocks_override : vs_default_grid(ctx->n_cu, m->n_slabs);
    else if (pl.variant == CANOPY_KERNEL_FILTER_REFINE || pl.variant == CANOPY_KERNEL_FIXED_SUM)
        pl.grid = fr_refine_grid(ctx->n_cu);
    else
        pl.grid = ctx->n_blocks_override > 0 ? ctx->n_blocks_override : rows_kernel_default_grid(ctx->n_cu, pl.variant, m->n_rows);
    pl.grid_fx = pl.variant == CANOPY_KERNEL_FIXED_SUM ? fx_sum_grid(ctx->n_cu, m->n_events, m->n_fr_slabs) : 0;
    pl.two_step = pl.variant == CANOPY_KERNEL_FILTER_REFINE || pl.variant == CANOPY_KERNEL_FIXED_SUM;
    pl.pipelined = ctx->pipeline_on && pl.two_step && !pl.want_mask && !pl.want_imp && !pl.want_rows && !ctx->capturing &&
                   !pl.filter_only && !(ctx->comm && ctx->comm->deferred) && !(ctx->d_gather != nullptr && ctx->comm == nullptr);
    int slot = -1; // the model's own set this pass reads (-1: an external one)
    for (int k = 0; k < kPSets; ++k)
        if (P == &m->psets[k]) slot = k;
    return launch_pass(ctx, m, P, slot, params, pl, opt);
}

} // namespace canopy

/*
 * canopy_oracle.h -- CPU oracle for the cut-set quantification path.
 *
 * TEST INFRASTRUCTURE ONLY. Nothing under canopy_amd/ or include/ may include, link, import or
 * execute anything in oracle/. Allowed callers: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py -- there only as the checker / reported baseline.
 *
 * PARITY UNPINNED BY THE REFERENCE: openpra-org/canopy holds no quantifier
 * (reference src/canopy.cpp:1-31 is a licence comment; include/canopy.h:33-36 an empty guard;
 * tests/, benchmarks/, examples/ are 0-byte CMakeLists.txt), so no golden vector, KAT or fixture
 * of the reference exists for this path and nothing can be compiled into oracle/_ref.
 * The only inputs the reference defines are the probability accessor
 * (src/mef/openpsa/event/event.h:93-96 BasicEvent::p()) and the knob names / defaults
 * (src/mef/openpsa/settings.h:19 Approximation{kRareEvent,kMcub}; :323 limit_order 20;
 * :332 cut_off 1e-8). This oracle is therefore the build's own statement of the canonical
 * arithmetic (include/canopy.h header comment), pinned instead by closed-form known answers and
 * a 50-digit mpmath cross-check (tests/golden/, tests/test_oracle.py).
 */
#ifndef CANOPY_ORACLE_H
#define CANOPY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_result {
    /* partial record, same meaning as canopy_partial (include/canopy.h) */
    double rare_hi, rare_lo;
    double log1p_hi, log1p_lo;
    double n_kept;
    double n_certain;
    double max_logp;
    double argmax_row;
    /* finished values */
    double rare_event; /* unclamped */
    double mcub;
    double elapsed_s;  /* wall time of the quantification loop only */
} oracle_result;

/*
 * Canonical quantification (settings.h:19 rare-event and MCUB, both accumulated).
 * row_ptr[0] must be 0. p[n_events] in [0,1]. cut_off <= 0 and limit_order == 0 disable the
 * filters. n_threads >= 1: contiguous nnz-balanced row shards, per-thread Neumaier sums,
 * combined in thread order. row_logp (n_rows doubles) and keep_mask (ceil(n_rows/64) words)
 * may be NULL. Returns 0, or -1 on invalid input.
 */
int oracle_quantify(int64_t n_rows, int32_t n_events, const int64_t* row_ptr,
                    const int32_t* col_idx, int64_t row_offset, const double* p, double cut_off,
                    int32_t limit_order, int32_t n_threads, oracle_result* out, double* row_logp,
                    uint64_t* keep_mask);

/* Product-domain cross-check: P_i = p[c0]*p[c1]*... (no logs), plain left-to-right sums. */
int oracle_quantify_product_domain(int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx,
                                   const double* p, double* rare, double* one_minus_prod);

/* Deterministic combine of shard records in index order (mirror of canopy_combine_partials). */
void oracle_combine(const oracle_result* parts, int32_t n, oracle_result* out);

/* ---- synthetic family (CPU restatement of canopy_model_synthetic) --------------------------- */
uint64_t oracle_draw(uint64_t seed, uint64_t stream, uint64_t index, uint64_t t);
int32_t oracle_synth_row_len(uint64_t seed, int64_t row, int32_t avg_nnz);
int64_t oracle_synth_nnz(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t avg_nnz);
/* row_ptr[n_rows+1] (rebased to 0) and col_idx[nnz]; returns 0 or -1 on invalid arguments. */
int oracle_synth_csr(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t n_events,
                     int32_t avg_nnz, int32_t n_threads, int64_t* row_ptr, int32_t* col_idx);
/* p_j = 10^(-8 + 5 u_j), u_j uniform [0,1): log-uniform on [1e-8, 1e-3). */
void oracle_synth_probabilities(uint64_t seed, int32_t n_events, double* p);

/*
 * K6 restatement: direct Monte-Carlo through the gate DAG (include/canopy.h, canopy_dag_view). Same
 * counter-based draw as the device: sample l of 64-sample word w of event j owns a 53-bit uniform u whose
 * bit (52 - k) is bit l of mix64(mix64(key ^ j) + 64 w + k); TRUE iff u < ceil(p_j * 2^53); key = stream key
 * of (seed, stream 4). Evaluated per sample, most significant bit first (the device resolves 64 samples at a
 * time with bit masks). Gates in the caller's (topological) order, one u64 word of 64 samples per node.
 * Returns hits, or -1.
 */
int64_t oracle_mc(int32_t n_events, int32_t n_gates, const int32_t* gate_type, const int32_t* gate_min,
                  const int32_t* arg_ptr, const int32_t* args, int32_t top, const double* p, uint64_t seed,
                  int64_t first_sample, int64_t num_trials, int32_t n_threads);

/*
 * Importance contributions (N4): contrib[j] = sum of P_i = exp(s_i) over the KEPT cut sets that contain basic
 * event j (same s_i, same filter as oracle_quantify). Fussell-Vesely importance is contrib[j] / rare_event,
 * the rare-event Birnbaum importance contrib[j] / p[j]. Accumulated in long double. Returns 0 or -1.
 */
int oracle_importance(int64_t n_rows, int32_t n_events, const int64_t* row_ptr, const int32_t* col_idx,
                      const double* p, double cut_off, int32_t limit_order, double* contrib);

/* nnz-balanced contiguous shard bounds (mirror of canopy_shard_rows). */
void oracle_shard_rows(const int64_t* row_ptr, int64_t n_rows, int32_t n_shards, int64_t* bounds);

#ifdef __cplusplus
}
#endif
#endif

/*
 * canopy_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see canopy_oracle.h).
 *
 * Parity unpinned by the reference: the reference has no quantifier to restate
 * (src/canopy.cpp:1-31). What is restated from it:
 *   - the input  p[j] = BasicEvent::p()           src/mef/openpsa/event/event.h:93-96
 *   - rare-event / MCUB as the two approximations  src/mef/openpsa/settings.h:19,22
 *   - cut_off (1e-8) and limit_order (20) filters  src/mef/openpsa/settings.h:323,332
 * The arithmetic itself is the canonical definition in include/canopy.h.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -shared -fPIC (see oracle/Makefile).
 * -ffp-contract=off keeps every a+b a plain IEEE add: s_i must be bit-identical to the
 * device's sequential v_add_f64 chain.
 */
#include "canopy_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- counter-based generator ---------------------------------------------------------------- */

static inline uint64_t mix64(uint64_t z) { /* splitmix64 step + finalizer */
    z += 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

enum { STREAM_LEN = 1, STREAM_COL = 2, STREAM_PROB = 3 };

uint64_t oracle_draw(uint64_t seed, uint64_t stream, uint64_t index, uint64_t t) {
    uint64_t key = mix64(seed ^ (stream * 0xD1B54A32D192ED03ULL));
    return mix64(mix64(key ^ index) + t);
}

/* uniform integer on [0, n) from the top 32 bits (n < 2^32) */
static inline uint32_t bounded(uint64_t h, uint32_t n) {
    return (uint32_t)(((h >> 32) * (uint64_t)n) >> 32);
}

int32_t oracle_synth_row_len(uint64_t seed, int64_t row, int32_t avg_nnz) {
    uint32_t span = (uint32_t)(2 * avg_nnz - 3); /* {2 .. 2*avg-2} */
    return 2 + (int32_t)bounded(oracle_draw(seed, STREAM_LEN, (uint64_t)row, 0), span);
}

int64_t oracle_synth_nnz(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t avg_nnz) {
    int64_t nnz = 0;
    for (int64_t i = 0; i < n_rows; ++i) nnz += oracle_synth_row_len(seed, first_row + i, avg_nnz);
    return nnz;
}

static void synth_row_cols(uint64_t seed, int64_t row, int32_t len, int32_t n_events, int32_t* out) {
    int32_t got = 0;
    for (uint64_t t = 0; got < len; ++t) {
        int32_t c = (int32_t)bounded(oracle_draw(seed, STREAM_COL, (uint64_t)row, t), (uint32_t)n_events);
        int dup = 0;
        for (int32_t k = 0; k < got; ++k) dup |= (out[k] == c);
        if (dup) continue; /* redraw on collision */
        /* insertion keeps the row ascending */
        int32_t k = got++;
        while (k > 0 && out[k - 1] > c) { out[k] = out[k - 1]; --k; }
        out[k] = c;
    }
}

typedef struct synth_job {
    uint64_t seed; int64_t first_row, r0, r1; int32_t n_events, avg_nnz;
    const int64_t* row_ptr; int32_t* col_idx;
} synth_job;

static void* synth_worker(void* arg) {
    synth_job* j = (synth_job*)arg;
    for (int64_t i = j->r0; i < j->r1; ++i) {
        int32_t len = (int32_t)(j->row_ptr[i + 1] - j->row_ptr[i]);
        synth_row_cols(j->seed, j->first_row + i, len, j->n_events, j->col_idx + j->row_ptr[i]);
    }
    return NULL;
}

int oracle_synth_csr(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t n_events,
                     int32_t avg_nnz, int32_t n_threads, int64_t* row_ptr, int32_t* col_idx) {
    if (n_rows < 0 || avg_nnz < 2 || avg_nnz > 17 || n_events < 2 * avg_nnz - 2) return -1;
    row_ptr[0] = 0;
    for (int64_t i = 0; i < n_rows; ++i)
        row_ptr[i + 1] = row_ptr[i] + oracle_synth_row_len(seed, first_row + i, avg_nnz);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    synth_job jobs[256];
    for (int32_t t = 0; t < n_threads; ++t) {
        jobs[t] = (synth_job){seed, first_row, n_rows * t / n_threads, n_rows * (t + 1) / n_threads,
                              n_events, avg_nnz, row_ptr, col_idx};
        if (t + 1 < n_threads) pthread_create(&th[t], NULL, synth_worker, &jobs[t]);
    }
    synth_worker(&jobs[n_threads - 1]);
    for (int32_t t = 0; t + 1 < n_threads; ++t) pthread_join(th[t], NULL);
    return 0;
}

void oracle_synth_probabilities(uint64_t seed, int32_t n_events, double* p) {
    for (int32_t j = 0; j < n_events; ++j) {
        double u = (double)(oracle_draw(seed, STREAM_PROB, (uint64_t)j, 0) >> 11) * 0x1.0p-53;
        p[j] = pow(10.0, -8.0 + 5.0 * u);
    }
}

/* ---- sharding --------------------------------------------------------------------------------- */

void oracle_shard_rows(const int64_t* row_ptr, int64_t n_rows, int32_t n_shards, int64_t* bounds) {
    int64_t base = row_ptr[0], nnz = row_ptr[n_rows] - base;
    bounds[0] = 0;
    for (int32_t k = 1; k < n_shards; ++k) {
        if (nnz == 0) { bounds[k] = n_rows * k / n_shards; continue; }
        int64_t target = base + (int64_t)(((__int128)nnz * k) / n_shards);
        int64_t lo = 0, hi = n_rows; /* first r with row_ptr[r] >= target */
        while (lo < hi) {
            int64_t mid = lo + (hi - lo) / 2;
            if (row_ptr[mid] >= target) hi = mid; else lo = mid + 1;
        }
        bounds[k] = lo < bounds[k - 1] ? bounds[k - 1] : lo;
    }
    bounds[n_shards] = n_rows;
}

/* ---- compensated sums ------------------------------------------------------------------------ */

typedef struct csum { double s, c; } csum;

static inline void csum_add(csum* a, double x) { /* Neumaier */
    double t = a->s + x;
    if (fabs(a->s) >= fabs(x)) a->c += (a->s - t) + x; else a->c += (x - t) + a->s;
    a->s = t;
}

static inline void csum_merge(csum* a, const csum* b) {
    csum_add(a, b->s);
    a->c += b->c;
}

/* renormalise so that hi + lo is the value and |lo| <= ulp(hi)/2 */
static inline void csum_split(const csum* a, double* hi, double* lo) {
    double h = a->s + a->c;
    *lo = (a->s - h) + a->c;
    *hi = h;
}

/* ---- quantification -------------------------------------------------------------------------- */

typedef struct quant_job {
    int64_t r0, r1, row_offset;
    const int64_t* row_ptr; const int32_t* col_idx; const double* logp;
    double log_cut; int32_t limit_order;
    double* row_logp; uint64_t* keep_mask;
    csum rare, l1p; int64_t n_kept, n_certain; double max_logp; int64_t argmax;
} quant_job;

static void* quant_worker(void* arg) {
    quant_job* j = (quant_job*)arg;
    csum rare = {0.0, 0.0}, l1p = {0.0, 0.0};
    int64_t n_kept = 0, n_certain = 0, argmax = -1;
    double max_logp = -INFINITY;
    for (int64_t i = j->r0; i < j->r1; ++i) {
        int64_t b = j->row_ptr[i], e = j->row_ptr[i + 1];
        double s = 0.0;
        for (int64_t k = b; k < e; ++k) s = s + j->logp[j->col_idx[k]]; /* CSR order, plain adds */
        int keep = (s >= j->log_cut) && (j->limit_order == 0 || (e - b) <= j->limit_order);
        if (j->row_logp) j->row_logp[i] = s;
        if (j->keep_mask && keep) j->keep_mask[i >> 6] |= (uint64_t)1 << (i & 63);
        if (!keep) continue;
        double P = exp(s);
        csum_add(&rare, P);
        if (P < 1.0) csum_add(&l1p, log1p(-P)); else ++n_certain;
        ++n_kept;
        if (s > max_logp) { max_logp = s; argmax = j->row_offset + i; }
    }
    j->rare = rare; j->l1p = l1p; j->n_kept = n_kept; j->n_certain = n_certain;
    j->max_logp = max_logp; j->argmax = argmax;
    return NULL;
}

static void finish(oracle_result* o) {
    o->rare_event = o->rare_hi + o->rare_lo;
    o->mcub = (o->n_certain > 0.0) ? 1.0 : -expm1(o->log1p_hi + o->log1p_lo);
}

int oracle_quantify(int64_t n_rows, int32_t n_events, const int64_t* row_ptr,
                    const int32_t* col_idx, int64_t row_offset, const double* p, double cut_off,
                    int32_t limit_order, int32_t n_threads, oracle_result* out, double* row_logp,
                    uint64_t* keep_mask) {
    if (n_rows < 0 || n_events < 0 || !row_ptr || row_ptr[0] != 0 || !out || limit_order < 0) return -1;
    for (int32_t jx = 0; jx < n_events; ++jx)
        if (!(p[jx] >= 0.0 && p[jx] <= 1.0)) return -1;
    double* logp = (double*)malloc(sizeof(double) * (size_t)(n_events > 0 ? n_events : 1));
    if (!logp) return -1;
    for (int32_t jx = 0; jx < n_events; ++jx) logp[jx] = log(p[jx]); /* once, on the host */
    double log_cut = cut_off > 0.0 ? log(cut_off) : -INFINITY;
    if (keep_mask) memset(keep_mask, 0, sizeof(uint64_t) * (size_t)((n_rows + 63) / 64));

    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    /* shard boundaries are multiples of 64 rows so that threads never share a mask word */
    int64_t bounds[257];
    oracle_shard_rows(row_ptr, n_rows, n_threads, bounds);
    for (int32_t t = 1; t < n_threads; ++t) {
        bounds[t] = (bounds[t] + 63) / 64 * 64;
        if (bounds[t] > n_rows) bounds[t] = n_rows;
        if (bounds[t] < bounds[t - 1]) bounds[t] = bounds[t - 1];
    }
    pthread_t th[256];
    quant_job* jobs = (quant_job*)calloc((size_t)n_threads, sizeof(quant_job));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int32_t t = 0; t < n_threads; ++t) {
        jobs[t].r0 = bounds[t]; jobs[t].r1 = bounds[t + 1]; jobs[t].row_offset = row_offset;
        jobs[t].row_ptr = row_ptr; jobs[t].col_idx = col_idx; jobs[t].logp = logp;
        jobs[t].log_cut = log_cut; jobs[t].limit_order = limit_order;
        jobs[t].row_logp = row_logp; jobs[t].keep_mask = keep_mask;
        if (t + 1 < n_threads) pthread_create(&th[t], NULL, quant_worker, &jobs[t]);
    }
    quant_worker(&jobs[n_threads - 1]);
    for (int32_t t = 0; t + 1 < n_threads; ++t) pthread_join(th[t], NULL);

    csum rare = {0.0, 0.0}, l1p = {0.0, 0.0};
    int64_t n_kept = 0, n_certain = 0, argmax = -1;
    double max_logp = -INFINITY;
    for (int32_t t = 0; t < n_threads; ++t) { /* thread order: deterministic */
        csum_merge(&rare, &jobs[t].rare);
        csum_merge(&l1p, &jobs[t].l1p);
        n_kept += jobs[t].n_kept; n_certain += jobs[t].n_certain;
        if (jobs[t].max_logp > max_logp) { max_logp = jobs[t].max_logp; argmax = jobs[t].argmax; }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    csum_split(&rare, &out->rare_hi, &out->rare_lo);
    csum_split(&l1p, &out->log1p_hi, &out->log1p_lo);
    out->n_kept = (double)n_kept; out->n_certain = (double)n_certain;
    out->max_logp = max_logp; out->argmax_row = (double)argmax;
    out->elapsed_s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    finish(out);
    free(jobs); free(logp);
    return 0;
}

void oracle_combine(const oracle_result* parts, int32_t n, oracle_result* out) {
    csum rare = {0.0, 0.0}, l1p = {0.0, 0.0};
    double n_kept = 0.0, n_certain = 0.0, max_logp = -INFINITY, argmax = -1.0;
    for (int32_t k = 0; k < n; ++k) {
        csum_add(&rare, parts[k].rare_hi); csum_add(&rare, parts[k].rare_lo);
        csum_add(&l1p, parts[k].log1p_hi); csum_add(&l1p, parts[k].log1p_lo);
        n_kept += parts[k].n_kept; n_certain += parts[k].n_certain;
        if (parts[k].n_kept > 0.0 && parts[k].argmax_row >= 0.0 &&
            (parts[k].max_logp > max_logp ||
             (parts[k].max_logp == max_logp && (argmax < 0.0 || parts[k].argmax_row < argmax)))) {
            max_logp = parts[k].max_logp; argmax = parts[k].argmax_row;
        }
    }
    memset(out, 0, sizeof(*out));
    csum_split(&rare, &out->rare_hi, &out->rare_lo);
    csum_split(&l1p, &out->log1p_hi, &out->log1p_lo);
    out->n_kept = n_kept; out->n_certain = n_certain; out->max_logp = max_logp; out->argmax_row = argmax;
    finish(out);
}

int oracle_quantify_product_domain(int64_t n_rows, const int64_t* row_ptr, const int32_t* col_idx,
                                   const double* p, double* rare, double* one_minus_prod) {
    long double r = 0.0L, prod = 1.0L;
    for (int64_t i = 0; i < n_rows; ++i) {
        long double P = 1.0L;
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) P *= (long double)p[col_idx[k]];
        r += P;
        prod *= (1.0L - P);
    }
    *rare = (double)r;
    *one_minus_prod = (double)(1.0L - prod);
    return 0;
}

int oracle_importance(int64_t n_rows, int32_t n_events, const int64_t* row_ptr, const int32_t* col_idx,
                      const double* p, double cut_off, int32_t limit_order, double* contrib) {
    if (n_rows < 0 || n_events < 0 || !row_ptr || !contrib || limit_order < 0) return -1;
    double* logp = (double*)malloc(sizeof(double) * (size_t)(n_events > 0 ? n_events : 1));
    long double* acc = (long double*)calloc((size_t)(n_events > 0 ? n_events : 1), sizeof(long double));
    if (!logp || !acc) { free(logp); free(acc); return -1; }
    for (int32_t jx = 0; jx < n_events; ++jx) {
        if (!(p[jx] >= 0.0 && p[jx] <= 1.0)) { free(logp); free(acc); return -1; }
        logp[jx] = log(p[jx]);
    }
    const double log_cut = cut_off > 0.0 ? log(cut_off) : -INFINITY;
    for (int64_t i = 0; i < n_rows; ++i) {
        double s = 0.0;
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) s = s + logp[col_idx[k]]; /* CSR order */
        const int64_t len = row_ptr[i + 1] - row_ptr[i];
        if (!(s >= log_cut) || (limit_order != 0 && len > limit_order)) continue;
        const long double P = (long double)exp(s);
        for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) acc[col_idx[k]] += P;
    }
    for (int32_t jx = 0; jx < n_events; ++jx) contrib[jx] = (double)acc[jx];
    free(logp); free(acc);
    return 0;
}

/* ---- K6: Monte-Carlo through the gate DAG ------------------------------------------------------- */

typedef struct mc_job {
    int32_t n_events, n_gates; const int32_t *gate_type, *gate_min, *arg_ptr, *args; int32_t top;
    const uint64_t* thresh; uint64_t key; int64_t first_word, w0, w1, num_trials; int64_t hits;
} mc_job;

static inline uint64_t mc_node(const uint64_t* node, int32_t a) { return a >= 0 ? node[a] : ~node[~a]; }

static void* mc_worker(void* arg) {
    mc_job* j = (mc_job*)arg;
    uint64_t* node = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(j->n_events + j->n_gates + 1));
    int64_t hits = 0;
    for (int64_t w = j->w0; w < j->w1; ++w) {
        for (int32_t ev = 0; ev < j->n_events; ++ev) {
            /* Sample l of this 64-sample word owns a 53-bit uniform u_l whose bit (52 - k) is bit l of the
               "plane" R_k = mix64(ek + 64 * word_index + k), k = 0 (most significant) .. 52; the sample is
               TRUE iff u_l < thresh (thresh = ceil(p * 2^53) <= 2^53). Evaluated here per sample, most
               significant bit first, planes hashed on demand (the GPU kernel resolves all 64 samples of a
               plane at once with bit masks: a different evaluation of the same definition). */
            const uint64_t ek = mix64(j->key ^ (uint64_t)ev), T = j->thresh[ev];
            const uint64_t base = ek + (uint64_t)(j->first_word + w) * 64u;
            uint64_t plane[53], word = 0;
            int n_planes = 0;
            if (T >> 53) word = ~(uint64_t)0; /* p == 1: every u < 2^53 */
            else for (int l = 0; l < 64; ++l) {
                for (int k = 0; k < 53; ++k) {
                    if (k == n_planes) { plane[k] = mix64(base + (uint64_t)k); ++n_planes; }
                    const unsigned ub = (unsigned)((plane[k] >> l) & 1u), tb = (unsigned)((T >> (52 - k)) & 1u);
                    if (ub != tb) { if (ub < tb) word |= (uint64_t)1 << l; break; }
                } /* all 53 bits equal: u == thresh, not below it */
            }
            node[ev] = word;
        }
        for (int32_t g = 0; g < j->n_gates; ++g) {
            int32_t b = j->arg_ptr[g], e = j->arg_ptr[g + 1], type = j->gate_type[g];
            uint64_t r = 0;
            if (type == 0 || type == 5) { r = ~(uint64_t)0; for (int32_t k = b; k < e; ++k) r &= mc_node(node, j->args[k]); if (type == 5) r = ~r; }
            else if (type == 1 || type == 6) { for (int32_t k = b; k < e; ++k) r |= mc_node(node, j->args[k]); if (type == 6) r = ~r; }
            else if (type == 3) { for (int32_t k = b; k < e; ++k) r ^= mc_node(node, j->args[k]); }
            else if (type == 4) r = ~mc_node(node, j->args[b]);
            else if (type == 7) r = mc_node(node, j->args[b]);
            else { /* ATLEAST: per sample, count TRUE arguments and compare with K (plain counting) */
                for (int l = 0; l < 64; ++l) {
                    int cnt = 0;
                    for (int32_t k = b; k < e; ++k) cnt += (int)((mc_node(node, j->args[k]) >> l) & 1u);
                    if (cnt >= j->gate_min[g]) r |= (uint64_t)1 << l;
                }
            }
            node[j->n_events + g] = r;
        }
        int64_t left = j->num_trials - w * 64;
        uint64_t valid = left >= 64 ? ~(uint64_t)0 : (((uint64_t)1 << left) - 1);
        hits += __builtin_popcountll(node[j->top] & valid);
    }
    free(node);
    j->hits = hits;
    return NULL;
}

int64_t oracle_mc(int32_t n_events, int32_t n_gates, const int32_t* gate_type, const int32_t* gate_min,
                  const int32_t* arg_ptr, const int32_t* args, int32_t top, const double* p, uint64_t seed,
                  int64_t first_sample, int64_t num_trials, int32_t n_threads) {
    if (n_events < 0 || n_gates < 0 || num_trials < 0 || first_sample < 0 || (first_sample & 63)) return -1;
    uint64_t* thresh = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(n_events + 1));
    for (int32_t j = 0; j < n_events; ++j) {
        if (!(p[j] >= 0.0 && p[j] <= 1.0)) { free(thresh); return -1; }
        thresh[j] = (uint64_t)ceil(p[j] * 9007199254740992.0);
    }
    uint64_t key = mix64(seed ^ ((uint64_t)4 * 0xD1B54A32D192ED03ULL));
    int64_t n_words = (num_trials + 63) / 64;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    mc_job jobs[256];
    for (int32_t t = 0; t < n_threads; ++t) {
        jobs[t] = (mc_job){n_events, n_gates, gate_type, gate_min, arg_ptr, args, top, thresh, key,
                           first_sample / 64, n_words * t / n_threads, n_words * (t + 1) / n_threads, num_trials, 0};
        if (t + 1 < n_threads) pthread_create(&th[t], NULL, mc_worker, &jobs[t]);
    }
    mc_worker(&jobs[n_threads - 1]);
    int64_t hits = jobs[n_threads - 1].hits;
    for (int32_t t = 0; t + 1 < n_threads; ++t) { pthread_join(th[t], NULL); hits += jobs[t].hits; }
    free(thresh);
    return hits;
}

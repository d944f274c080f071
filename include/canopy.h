/*
 * canopy.h -- C ABI of the MI355X-native cut-set quantification path.
 *
 * The reference ships this header as an empty include guard
 * (reference include/canopy.h:33-36) and src/canopy.cpp as a licence comment
 * (reference src/canopy.cpp:1-31): the quantifier is the hole this library fills.
 * Conventions kept from the reference scaffold:
 *   - plain C header, no C++ in the signatures (cmake/canopy-config.cmake.in:25
 *     promises C consumers; int64_t needs <stdint.h>, so C99 is the floor),
 *   - CANOPY_SHARED / CANOPY_BUILD visibility defines (reference src/CMakeLists.txt:19-23),
 *   - a separate C++-only RAII header (reference include/canopy-cpp.h:34-40).
 * Parameter names and defaults mirror mef::openpsa::Settings
 * (reference src/mef/openpsa/settings.h:19,22,314-332): approximation
 * {rare-event, mcub, monte-carlo}, cut_off 1e-8, limit_order 20, num_trials 1e3, seed 0.
 *
 * Semantics (the canonical arithmetic; oracle/canopy_oracle.c is its CPU restatement):
 *   rows    = minimal cut sets, columns = basic events, CSR, columns ascending per row
 *   logp[j] = log(p[j])                      computed once on the host (glibc log)
 *   s_i     = ((logp[c0] + logp[c1]) + ...)  plain fp64 adds in CSR order
 *   keep_i  = s_i >= log(cut_off)  &&  (limit_order == 0 || len_i <= limit_order)
 *   P_i     = exp(s_i)
 *   rare    = sum_{keep} P_i                 (compensated)
 *   mcub    = 1 - prod_{keep} (1 - P_i) = -expm1( sum_{keep} log1p(-P_i) )
 *   n_kept, argmax_i s_i (ties -> lowest i), optional keep bitmask and per-row s_i
 * Integer outputs are bit-exact against the oracle; rare/mcub agree to <= 1e-12 relative.
 *
 * Every entry point returns canopy_status; no exception crosses this boundary.
 * The library never falls back to a CPU path: without a usable HIP device
 * canopy_ctx_create fails with CANOPY_ERR_NO_DEVICE.
 * The library reads NO environment variable (the tuning switches of benchmarks/ exist only in a
 * -DCANOPY_ABLATIONS build).
 *
 * Threads. A canopy_ctx is bound to ONE device (one process per GPU) and is NOT thread-safe: one host
 * thread at a time per context; distinct contexts may be driven from distinct threads. A canopy_model
 * may be QUANTIFIED through several contexts of its device at once (also from several threads): a pass
 * takes the model as const -- the derived layouts and tables a pass may have to build on first use
 * are built once, under a mutex of the model. The calls that change a model
 * (canopy_model_set_probabilities / _set_log_probabilities / _prepare / _free,
 * canopy_quantify_batch*) must not overlap with any other call on the same model; when passes of
 * another context still read the model, set_probabilities waits for the whole device first.
 */
#ifndef CANOPY_CANOPY_H
#define CANOPY_CANOPY_H

#include <stddef.h>
#include <stdint.h>

#if defined(CANOPY_SHARED) && defined(_WIN32)
#  ifdef CANOPY_BUILD
#    define CANOPY_API __declspec(dllexport)
#  else
#    define CANOPY_API __declspec(dllimport)
#  endif
#elif defined(__GNUC__)
#  define CANOPY_API __attribute__((visibility("default")))
#else
#  define CANOPY_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define CANOPY_VERSION_MAJOR 0
#define CANOPY_VERSION_MINOR 1
#define CANOPY_VERSION_PATCH 0

/* ---- status ------------------------------------------------------------------------------- */
typedef enum canopy_status {
    CANOPY_OK = 0,
    CANOPY_ERR_INVALID_ARGUMENT = -1,
    CANOPY_ERR_OUT_OF_MEMORY = -2,
    CANOPY_ERR_HIP = -3,
    CANOPY_ERR_NO_DEVICE = -4,
    CANOPY_ERR_UNSUPPORTED = -5,
    CANOPY_ERR_INVALID_MODEL = -6, /* CSR failed validation (row_ptr not monotone, column out of
                                      range, columns not strictly ascending within a row) */
    CANOPY_ERR_STATE = -7,         /* call order violated (e.g. quantify before set_probabilities) */
    CANOPY_ERR_INTERNAL = -8,      /* an unexpected C++ exception was caught at the boundary */
    CANOPY_ERR_RCCL = -9           /* an RCCL call of the cross-GPU combine failed (canopy_last_error has its text) */
} canopy_status;

CANOPY_API const char* canopy_status_string(canopy_status s);
CANOPY_API void canopy_version(int* major, int* minor, int* patch);

/* ---- context ------------------------------------------------------------------------------ */
typedef struct canopy_ctx canopy_ctx;

typedef struct canopy_ctx_params {
    int32_t device;  /* HIP device ordinal; the process-per-GPU launcher passes LOCAL_RANK */
    int32_t n_blocks;/* persistent grid size override, 0 = library default for the device */
    void* stream;    /* hipStream_t to launch on, NULL = the library creates its own */
} canopy_ctx_params;

CANOPY_API canopy_status canopy_ctx_create(const canopy_ctx_params* params, canopy_ctx** out);
CANOPY_API void canopy_ctx_free(canopy_ctx* ctx);
/* Last HIP / validation error text captured by this context ("" if none). Owned by the ctx. */
CANOPY_API const char* canopy_last_error(const canopy_ctx* ctx);
/* Block the host until everything the context has enqueued (on its stream and on its side stream) has finished. */
CANOPY_API canopy_status canopy_ctx_synchronize(canopy_ctx* ctx);
/*
 * Pipelining of consecutive passes (default: on). A sum-only FILTER_REFINE / FIXED_SUM pass is a row kernel that streams the whole model (HBM-bound, every
 * CU busy) followed by a latency-bound tail (exact pass over the listed rows, final reduce, and with a communicator
 * the all-reduce). With pipelining on, consecutive such passes rotate through four internal lanes (canopy_ctx_set_pipeline's default) -- a stream, scratch,
 * record and all-reduce slot buffer each -- and the device overlaps them: the next pass's row kernel takes the CUs as
 * this pass's row kernel leaves them, while this pass's tail runs beside it. A sequence of passes (uncertainty trials,
 * repeated quantification after canopy_model_set_probabilities, the benchmark's steps) is then paced by the row kernel
 * alone. canopy_quantify_fetch / canopy_comm_fetch deliver the LAST enqueued pass and wait for every lane;
 * canopy_quantify_enqueue hands out a TICKET per pass instead, and canopy_ticket_fetch / canopy_comm_ticket_fetch wait
 * for that pass's lane only -- a caller that needs every pass's result (uncertainty trials, a new probability vector
 * per pass) keeps up to four passes in flight and still runs at the pipelined rate. canopy_model_set_probabilities
 * does not drain the lanes either (it writes the next of the model's four probability sets). The host
 * may block in canopy_quantify_async until the pass four back has finished; passes that ask for per-row outputs,
 * keep masks or importance, all other kernel variants, and contexts with a caller-owned gather buffer
 * (canopy_ctx_set_gather_buffer) run un-pipelined on the context's stream behind whatever is still in flight.
 * enable: 0 = off (one pass at a time on the context's stream), 1 = on with the default of four lanes, 2 .. 4 = on with that
 * many lanes (a fourth lane buys slack for a slow cross-GPU collective in the tail of a short shard pass). The call
 * waits for everything in flight.
 */
CANOPY_API canopy_status canopy_ctx_set_pipeline(canopy_ctx* ctx, int32_t enable);

/* ---- model: the CSR cut-set x basic-event incidence matrix, resident in HBM ----------------- */
typedef struct canopy_model canopy_model;

typedef struct canopy_csr_view {
    int64_t n_rows;         /* R: cut sets held by THIS shard */
    int32_t n_events;       /* E: basic events (columns) */
    int32_t reserved;
    const int64_t* row_ptr; /* [R+1], row_ptr[0] may be non-zero (a slice of a larger array) */
    const int32_t* col_idx; /* [row_ptr[R]-row_ptr[0]] when sliced: entry k of the view is
                               col_idx[k - row_ptr[0]] ... see canopy_model_upload */
    int64_t row_offset;     /* global index of this shard's row 0 (argmax is reported globally) */
} canopy_csr_view;

enum {
    CANOPY_UPLOAD_VALIDATE = 1u,      /* run the CSR validation kernel (default for host callers) */
    CANOPY_UPLOAD_DEVICE_POINTERS = 2u /* row_ptr/col_idx already live in device memory; they are
                                          copied device-to-device (the library owns its copy) */
};

/*
 * Copies the CSR to the device. row_ptr is rebased so that the stored row_ptr[0] == 0;
 * col_idx must point at the entry that row_ptr[0] designates (i.e. the slice's first entry).
 * The caller keeps ownership of the host arrays and may free them when this returns.
 */
CANOPY_API canopy_status canopy_model_upload(canopy_ctx* ctx, const canopy_csr_view* csr,
                                             uint32_t flags, canopy_model** out);

/*
 * Generates, directly in HBM, rows [first_row, first_row + n_rows) of the deterministic
 * synthetic CSR family of the benchmark driver (replaces the reference's empty benchmarks/,
 * reference benchmarks/CMakeLists.txt): row length uniform on {2 .. 2*avg_nnz-2},
 * len distinct uniform columns per row, ascending. Counter-based (splitmix64), so any
 * row range of the same (seed, n_events, avg_nnz) family is identical wherever it is made;
 * oracle/canopy_oracle.c holds the CPU restatement. row_offset of the model = first_row.
 */
CANOPY_API canopy_status canopy_model_synthetic(canopy_ctx* ctx, uint64_t seed, int64_t first_row,
                                                int64_t n_rows, int32_t n_events, int32_t avg_nnz,
                                                canopy_model** out);

CANOPY_API void canopy_model_free(canopy_model* model);

typedef struct canopy_model_info {
    int64_t n_rows;
    int64_t nnz;
    int64_t row_offset;
    int32_t n_events;
    int32_t max_row_len;
    int64_t device_bytes;       /* bytes of HBM the model holds (canonical CSR + derived layouts) */
    int64_t stream_bytes;       /* bytes one pass over the model's derived stream reads (the slab stream in the form the
                                   filter reads -- 24-bit entries when the model allows --, else the value-scatter or
                                   wave-slab stream), 0 if none is built */
    int64_t algorithmic_bytes;  /* (R+1)*8 + nnz*4 + E*8: compulsory traffic of one sum-only pass */
    int64_t stream_bytes_wide;  /* bytes one pass over the 32-bit form of the slab stream reads (FIXED_SUM, the gathered filter:
                                   4 B per entry in 1 KiB chunks + chunk offsets + the fixed-point table once), 0 without a slab stream */
    int64_t stream_bytes_exact; /* bytes one pass of the exact all-rows stream kernel reads (the band stream when the model has
                                   one, else the wave-slab stream; entries + offsets + the logp table once), 0 if neither is built */
    int64_t stream_bytes_rows;  /* bytes the quantised filter of FILTER_REFINE reads when the model has a ROW STREAM (up to 260 096
                                   events: 12 .. 18 bits per entry in length-sorted rows, no row ids; range descriptors, item-end
                                   masks, the logp table once), 0 without one -- a sum-only FILTER_REFINE pass then reads
                                   stream_bytes. Its lists hold positions: the exact pass reads a position word per listed row
                                   (row id, row start, row length: the candidates of an item share its 128-byte line) */
} canopy_model_info;

CANOPY_API canopy_status canopy_model_get_info(const canopy_model* model, canopy_model_info* info);

/* Copy the device CSR back (row_ptr rebased to 0). Either pointer may be NULL. */
CANOPY_API canopy_status canopy_model_download(canopy_ctx* ctx, const canopy_model* model,
                                               int64_t* row_ptr, int32_t* col_idx);

/*
 * Basic-event probabilities (reference BasicEvent::p(), src/mef/openpsa/event/event.h:93-96),
 * p[E] on the host, each in [0,1]. logp = log(p) is evaluated here, on the host, once
 * (canonical semantics above) and uploaded; quantify calls then run with inputs resident.
 */
CANOPY_API canopy_status canopy_model_set_probabilities(canopy_ctx* ctx, canopy_model* model,
                                                        const double* p);
/*
 * The same with the natural logarithms supplied by the caller (logp[E], each <= 0, -inf for p = 0): no host log.
 * A caller that wants results identical to canopy_model_set_probabilities passes glibc's log(p[j]).
 *
 * Both calls are ASYNCHRONOUS towards the device and drain nothing: the values go (host log on the library's worker
 * threads, then one copy from a pinned staging buffer) into the NEXT of four probability sets the model keeps; passes
 * already enqueued keep reading the set they were enqueued with, the passes enqueued afterwards read the new one. The
 * host array may be reused as soon as the call returns. Only while a captured pass (canopy_graph_capture) of the
 * model exists is the current set rewritten in place, behind everything in flight.
 */
CANOPY_API canopy_status canopy_model_set_log_probabilities(canopy_ctx* ctx, canopy_model* model,
                                                            const double* logp);

/*
 * Build the derived device layout a kernel variant needs: the wave-slab stream (WAVE_STREAM), the band stream (BAND_STREAM),
 * the value-scatter stream (VALUE_SCATTER), for the two slab passes the chunk offsets, the exact pass's row offsets and -- up
 * to 260 096 events -- the quantised filter's ROW STREAM (FILTER_REFINE; FIXED_SUM also builds the column-sorted chunks it
 * reads, which a model with a row stream otherwise leaves to the first pass that needs them); a no-op for the row kernels.
 * One-off per model, independent of p.
 * kernel = CANOPY_KERNEL_AUTO prepares what AUTO will pick. canopy_quantify_async prepares lazily
 * (and synchronously) when this was not called (canopy_graph_capture runs one ordinary pass before it captures, for that reason). Returns CANOPY_ERR_UNSUPPORTED if the variant was
 * asked for explicitly and the model cannot be represented in it (E >= 2^19 or absurd row lengths).
 */
CANOPY_API canopy_status canopy_model_prepare(canopy_ctx* ctx, canopy_model* model, int32_t kernel);

/* ---- quantification ------------------------------------------------------------------------ */
typedef enum canopy_approximation { /* reference settings.h:19 Approximation */
    CANOPY_APPROX_RARE_EVENT = 1,
    CANOPY_APPROX_MCUB = 2,
    CANOPY_APPROX_BOTH = 3
} canopy_approximation;

enum {
    CANOPY_QUANT_CLAMP = 1u,       /* rare_event = min(1, sum) */
    CANOPY_QUANT_ROW_LOGP = 2u,    /* also write s_i for every row (R*8 B of extra HBM stores) */
    CANOPY_QUANT_KEEP_MASK = 4u,   /* also write the survivor bitmask, bit (i%64) of word i/64 */
    CANOPY_QUANT_IMPORTANCE = 8u   /* also accumulate the per-event importance contributions, see
                                      canopy_fetch_importance (one extra pass over the kept rows) */
};

typedef struct canopy_quant_params {
    int32_t approximation; /* canopy_approximation; both sums are always accumulated */
    int32_t limit_order;   /* reference settings.h:323 default 20; 0 disables */
    double cut_off;        /* reference settings.h:332 default 1e-8; 0 disables */
    uint32_t flags;
    int32_t kernel;        /* 0 = auto; otherwise force a CANOPY_KERNEL_* variant */
} canopy_quant_params;

enum {
    CANOPY_KERNEL_AUTO = 0,
    CANOPY_KERNEL_ROWS = 1,     /* lane-per-row, LDS-staged col_idx slab, logp gathered via L2 */
    CANOPY_KERNEL_ROWS_LDS = 2, /* same, logp table resident in LDS (needs E*8 B <= table budget) */
    CANOPY_KERNEL_WAVE_STREAM = 3, /* derived layout: per-wave column-ordered entry stream, row sums in
                                      LDS (large E); built once by canopy_model_prepare */
    CANOPY_KERNEL_VALUE_SCATTER = 4, /* derived layout: per-slab column-ordered entry stream, gathered
                                        values scattered to CSR slots in LDS, lane-per-row sums (large E) */
    CANOPY_KERNEL_FILTER_REFINE = 5, /* cut_off > 0, no per-row output: a fast unordered pass flags the rows
                                       that might survive (up to 260 096 events: integer sums of an 8- or 4-bit table in LDS over
                                       a stream of length-sorted rows, 12 .. 18 bits per entry), an exact CSR-order pass recomputes
                                       only those. Reported values are exactly those of the other variants. */
    CANOPY_KERNEL_FIXED_SUM = 6,    /* no per-row output, rows of <= 63 entries: every row summed in 2^-47 fixed point
                                       (integer adds commute, so all waves share a slab: no ordering constraint),
                                       exp / log1p taken from that sum (|error| <= L*2^-48 + L*|s|*2^-53, inside the
                                       1e-12 contract); the rows whose keep decision or argmax rank that error could
                                       change are recomputed in exact CSR order. Integer outputs are the canonical ones
                                       bit for bit; AUTO's choice for cut_off = 0 and for cut-offs most rows survive. */
    CANOPY_KERNEL_BAND_STREAM = 7   /* derived layout: slabs of 16384 rows shared by the 16 waves of a workgroup, entries in
                                       column-sorted BANDS in which a row appears at most once, a workgroup barrier between
                                       bands: exact CSR-order sums (per-row output, keep mask, importance) with four times
                                       denser table gathers than WAVE_STREAM. E <= 131072; AUTO's choice there from 2e6
                                       rows when per-row outputs are asked for (WAVE_STREAM for a model it does not fit). */
};

CANOPY_API void canopy_quant_params_default(canopy_quant_params* params);

/*
 * One shard's partial result. Fixed layout of CANOPY_PARTIAL_DOUBLES doubles so that the
 * cross-GPU combine is a single sum all-reduce of world_size records (each rank fills only
 * its own slot), then canopy_combine_partials in rank order: deterministic on every rank.
 */
#define CANOPY_PARTIAL_DOUBLES 8
typedef struct canopy_partial {
    double rare_hi;     /* compensated sum of P_i over kept rows: value = hi + lo */
    double rare_lo;
    double log1p_hi;    /* compensated sum of log1p(-P_i) over kept rows with P_i < 1 */
    double log1p_lo;
    double n_kept;      /* exact integer < 2^53 */
    double n_certain;   /* kept rows with P_i >= 1 (mcub = 1 if any) */
    double max_logp;    /* max s_i over kept rows, -inf if none */
    double argmax_row;  /* global row index of that max (lowest on ties), -1 if none */
} canopy_partial;

typedef struct canopy_quant_result {
    double rare_event;
    double mcub;
    int64_t n_kept;
    int64_t argmax_row;
    double max_logp;
    double elapsed_kernel_ms; /* device time of the last pass (HIP events), -1 if not timed */
} canopy_quant_result;

/*
 * Enqueue one quantification pass over the model on the context's stream: the row kernel
 * plus the deterministic final reduce. Returns without waiting. The partial record lands in
 * the context's device buffer (canopy_ctx_partials_device).
 */
CANOPY_API canopy_status canopy_quantify_async(canopy_ctx* ctx, const canopy_model* model,
                                               const canopy_quant_params* params);
/* Wait for everything the context has enqueued, copy the partial record of the LAST pass to the host. */
CANOPY_API canopy_status canopy_quantify_fetch(canopy_ctx* ctx, canopy_partial* out);

/*
 * Tickets: one result per pass without serialising the passes. canopy_quantify_enqueue is canopy_quantify_async plus a
 * ticket (the pass's number on this context, never 0); the pass's final reduce also writes its record into pinned host
 * memory -- slot (ticket mod 64) of a ring -- with an event of its own behind the pass, so canopy_ticket_fetch is a wait
 * for THAT pass and a memcpy: no lane is joined, no device copy is enqueued. A ticket stays fetchable, in any order and
 * any number of times, until 64 later passes (ticketed or not, pipelined or not) have been enqueued on the context;
 * after that, and after a communicator has been attached to or freed from the context, it gives CANOPY_ERR_STATE.
 * (How many passes are IN FLIGHT is bounded by the lanes, not by the ring: enqueue blocks the host while the lane it
 * needs is still busy.) canopy_ticket_done polls without blocking. With a communicator attached
 * (canopy_comm_init_rank) canopy_comm_ticket_fetch delivers the COMBINED result of the pass.
 */
typedef uint64_t canopy_ticket;
CANOPY_API canopy_status canopy_quantify_enqueue(canopy_ctx* ctx, const canopy_model* model,
                                                 const canopy_quant_params* params, canopy_ticket* ticket);
CANOPY_API canopy_status canopy_ticket_fetch(canopy_ctx* ctx, canopy_ticket ticket, canopy_partial* out);
CANOPY_API canopy_status canopy_ticket_done(canopy_ctx* ctx, canopy_ticket ticket, int32_t* done);
/* Synchronous convenience: async + fetch + combine of the single shard; fills elapsed_kernel_ms. */
CANOPY_API canopy_status canopy_quantify(canopy_ctx* ctx, const canopy_model* model,
                                         const canopy_quant_params* params,
                                         canopy_quant_result* result);

/* Device address of the canopy_partial record of the last enqueued pass (CANOPY_PARTIAL_DOUBLES doubles). */
CANOPY_API void* canopy_ctx_partials_device(canopy_ctx* ctx);

/*
 * Multi-GPU (one process per GPU): point the context at a device buffer of
 * world_size * CANOPY_PARTIAL_DOUBLES doubles. Every later pass also writes this rank's record
 * into slot `rank` and zeroes the other slots in its final-reduce kernel, so that ONE
 * sum all-reduce (RCCL over xGMI) of the buffer delivers every rank's record to every rank --
 * no other collective, no host round trip. Pass NULL to detach.
 */
CANOPY_API canopy_status canopy_ctx_set_gather_buffer(canopy_ctx* ctx, void* device_buffer,
                                                      int32_t world_size, int32_t rank);

/*
 * K5 inside the library: the single RCCL all-reduce of the path, issued by libcanopy itself on the context's
 * stream right behind the final-reduce kernel -- a C or C++ host (the reference's library target,
 * src/CMakeLists.txt:9-23) needs nothing else for the multi-GPU path. The library owns the world x 8 slot buffer
 * (canopy_ctx_set_gather_buffer semantics); librccl.so.1 is bound on first use, so single-GPU hosts need no RCCL.
 *
 *   one process per GPU:   rank 0 calls canopy_comm_get_unique_id and hands the 128 bytes to the other ranks by
 *                          whatever channel the launcher has (MPI, a file, an environment variable, a TCP store);
 *                          EVERY rank then calls canopy_comm_init_rank (collective). From then on every
 *                          canopy_quantify_async on that context ends with the all-reduce, canopy_quantify returns
 *                          the COMBINED result on every rank (so every rank must call it), and
 *                          canopy_comm_fetch delivers the combined result of an asynchronous pass.
 *   one process, N GPUs:   canopy_comm_init_all over N contexts (one per device), then
 *                          canopy_comm_quantify_all_async runs the pass on every device and issues the N collectives
 *                          inside ncclGroupStart / ncclGroupEnd; canopy_comm_fetch on any of them.
 * The fold over the world records is done on the host in rank order: bitwise identical on every rank.
 * Monte-Carlo (K6): with a communicator attached (process-per-GPU shape) canopy_mc_run_async also sums
 * {hits, trials} over the ranks (one u64 all-reduce); canopy_comm_mc_fetch returns the global estimate.
 * canopy_comm_free detaches (the context returns to single-shard behaviour); free it before its context. (If the context
 * goes first, canopy_ctx_free detaches the communicator: every call on it then fails with CANOPY_ERR_INVALID_ARGUMENT except
 * canopy_comm_free, which releases it.)
 * Streams: the all-reduce is enqueued right behind the pass's final reduce, on the lane that ran it
 * (canopy_ctx_set_pipeline): the next pass's row kernel streams on another lane while the (latency-bound) collective
 * of this one is on the wire. Consecutive passes therefore issue the collectives of one communicator from different
 * streams; canopy_comm_set_mode picks how they are ordered on the device:
 *   CANOPY_COMM_CHAINED (default)  each collective waits for an event recorded behind the previous one: strict issue
 *                                  order on the device, the same on every rank, by construction (one event edge per
 *                                  pass, in the lane's latency-bound tail);
 *   CANOPY_COMM_UNORDERED          no edge: RCCL chains a communicator's launches itself (saves the edge; relies on
 *                                  RCCL's handling of one communicator on several streams);
 *   CANOPY_COMM_OWN_STREAM         every collective on ONE stream of the communicator's own, behind an event of its
 *                                  lane (two cross-queue edges per pass).
 * canopy_comm_fetch waits for the LAST enqueued pass; canopy_comm_ticket_fetch for the pass of a ticket only.
 */
#define CANOPY_COMM_ID_BYTES 128
typedef struct canopy_comm_id { char bytes[CANOPY_COMM_ID_BYTES]; } canopy_comm_id;
typedef struct canopy_comm canopy_comm;

CANOPY_API canopy_status canopy_comm_get_unique_id(canopy_comm_id* id);
CANOPY_API canopy_status canopy_comm_init_rank(canopy_ctx* ctx, const canopy_comm_id* id, int32_t world_size,
                                               int32_t rank, canopy_comm** out);
CANOPY_API canopy_status canopy_comm_init_all(canopy_ctx* const* ctxs, int32_t n, canopy_comm** out /* [n] */);
CANOPY_API void canopy_comm_free(canopy_comm* comm);
/* Any of the three may be NULL. rccl_version: NCCL-style code of the bound library (e.g. 22707). */
CANOPY_API canopy_status canopy_comm_info(const canopy_comm* comm, int32_t* world_size, int32_t* rank,
                                          int32_t* rccl_version);
CANOPY_API canopy_status canopy_comm_quantify_all_async(canopy_comm* const* comms, const canopy_model* const* models,
                                                        int32_t n, const canopy_quant_params* params);
enum { CANOPY_COMM_CHAINED = 0, CANOPY_COMM_UNORDERED = 1, CANOPY_COMM_OWN_STREAM = 2 };
/* Waits for everything in flight on the context, then switches. Every rank must use the same mode. */
CANOPY_API canopy_status canopy_comm_set_mode(canopy_comm* comm, int32_t mode);

/* Host-side, deterministic combine of n shard records in index order. */
CANOPY_API canopy_status canopy_combine_partials(const canopy_partial* parts, int32_t n,
                                                 uint32_t flags, canopy_quant_result* result);

/* Wait for the context's stream, copy the world records to the host, fold them in rank order. result and/or
   parts[world_size] may be NULL. flags: CANOPY_QUANT_CLAMP. */
CANOPY_API canopy_status canopy_comm_fetch(canopy_comm* comm, uint32_t flags, canopy_quant_result* result,
                                           canopy_partial* parts);
/* The same for the pass of a ticket (canopy_quantify_enqueue on the communicator's context): waits for that pass
   only; valid as long as the ticket is. Every rank holds the same records, so every rank gets the same result. */
CANOPY_API canopy_status canopy_comm_ticket_fetch(canopy_comm* comm, canopy_ticket ticket, uint32_t flags,
                                                  canopy_quant_result* result, canopy_partial* parts);

/* Device time of the last pass's all-reduce (event pair around it on the stream it ran on; waits for it). The pair is
   recorded for timed passes only -- the synchronous canopy_quantify, or a pass inside canopy_timing_begin/end --
   CANOPY_ERR_STATE otherwise (an event record costs microseconds a pass of an 8-way shard does not have). */
CANOPY_API canopy_status canopy_comm_last_collective_ms(canopy_comm* comm, double* ms);

/* Optional per-row outputs of the last pass that requested them (host destination). */
CANOPY_API canopy_status canopy_fetch_row_logp(canopy_ctx* ctx, const canopy_model* model,
                                               int64_t first_row, int64_t n, double* out);
CANOPY_API canopy_status canopy_fetch_keep_mask(canopy_ctx* ctx, const canopy_model* model,
                                                int64_t first_word, int64_t n_words, uint64_t* out);

/*
 * Importance contributions of the last pass that requested CANOPY_QUANT_IMPORTANCE:
 *   contrib[j] = sum of P_i over the KEPT cut sets that contain basic event j     (n_events doubles)
 * from which the usual measures follow on the host: Fussell-Vesely FV_j = contrib[j] / rare_event, rare-event
 * Birnbaum B_j = contrib[j] / p[j], risk reduction worth 1 / (1 - FV_j). (Importance analysis is a flag of the
 * reference's Settings, src/mef/openpsa/settings.h:262-279; the reference has no implementation.) The kept set and
 * every P_i are those of the pass itself; the sums are fp64 atomic adds in no fixed order, so they agree with the
 * oracle to <= 1e-12 relative but are not bitwise reproducible. Shards add: the cross-GPU combine is one sum
 * all-reduce of n_events doubles -- canopy_ctx_set_importance_buffer points the pass at a caller-owned device
 * buffer (e.g. a torch tensor handed to RCCL); NULL returns to the context's own buffer.
 */
CANOPY_API canopy_status canopy_fetch_importance(canopy_ctx* ctx, const canopy_model* model, double* contrib);
CANOPY_API canopy_status canopy_ctx_set_importance_buffer(canopy_ctx* ctx, void* device_buffer, int32_t n_events);

/*
 * Device time of the row kernel(s) and of the whole pass of the last synchronous canopy_quantify (HIP events; blocks
 * until they have completed). Asynchronous passes carry events only inside a timing region (canopy_timing_begin):
 * every event record is a marker packet between two kernels, microseconds of queue time on a small shard.
 */
CANOPY_API canopy_status canopy_last_kernel_ms(canopy_ctx* ctx, double* row_kernel_ms,
                                               double* total_ms);
/* The CANOPY_KERNEL_* variant the last canopy_quantify_async on this context actually ran (what AUTO
   resolved to); 0 before the first pass. */
CANOPY_API int32_t canopy_last_kernel_variant(const canopy_ctx* ctx);

/* FILTER_REFINE: rows the filter pass flagged for the exact pass, and the rows it looked at, as of the last
   canopy_quantify_fetch / canopy_quantify of such a pass (CANOPY_ERR_STATE before the first). */
CANOPY_API canopy_status canopy_last_filter_candidates(const canopy_ctx* ctx, int64_t* n_candidates, int64_t* n_rows);

/* ---- captured passes (hipGraph) -------------------------------------------------------------------------- */
/*
 * A small model is launch-bound (config 2: 1e5 cut sets, ~10 us of kernel time in 2-4 launches). canopy_graph_capture
 * records the launches canopy_quantify_async would enqueue for (model, params) into a HIP graph; canopy_graph_launch
 * replays them with ONE launch on the context's stream. Results are fetched as usual (canopy_quantify_fetch, the
 * gather buffer, canopy_fetch_keep_mask ...). The probabilities may be changed between launches
 * (canopy_model_set_probabilities writes the same device buffer); per-launch timing is not available for a replay.
 * The context, the model and its derived layouts must outlive the graph; capture once per (model, params).
 * A graph holds raw pointers to the context's scratch and output buffers as they were at capture. Anything that
 * replaces one of them afterwards -- an eager pass over a larger model, the first pass that asks for a per-row
 * output or importance, canopy_ctx_set_gather_buffer / canopy_ctx_set_importance_buffer with another pointer --
 * invalidates the graph: canopy_graph_launch then returns CANOPY_ERR_STATE and the pass must be captured again.
 */
typedef struct canopy_graph canopy_graph;
CANOPY_API canopy_status canopy_graph_capture(canopy_ctx* ctx, const canopy_model* model,
                                              const canopy_quant_params* params, canopy_graph** out);
CANOPY_API canopy_status canopy_graph_launch(canopy_ctx* ctx, canopy_graph* graph);
CANOPY_API void canopy_graph_free(canopy_graph* graph);

/* ---- N4: batched quantification -- K probability vectors ("trials") against one model ------------- */
/*
 * Reference vocabulary: uncertainty analysis quantifies the same cut sets under num_trials sampled
 * probability vectors (Settings::uncertainty_analysis / num_trials, src/mef/openpsa/settings.h:118-130,
 * 281-295; the deviates of src/mef/openpsa/expr/random_deviate.h draw them). The reference has no
 * implementation. Semantics here: p_trials is a host array [n_trials][n_events], row-major; the record of
 * trial k is exactly what canopy_quantify_async + canopy_quantify_fetch deliver after
 * canopy_model_set_probabilities(p_trials[k]) -- integer outputs bit-exact, sums to <= 1e-12 relative.
 * How: one ordinary pass under the element-wise maximum of the trials finds every row that can survive
 * the cut-off in ANY trial (IEEE addition is monotone, so this needs no tolerance); those rows are
 * compacted and each trial is evaluated exactly on the compact CSR, 8 trials per lane. The model's own
 * probabilities (canopy_model_set_probabilities) are neither needed nor changed.
 */
enum {
    CANOPY_BATCH_LOG_INPUT = 1u /* p_trials already holds natural logarithms (<= 0); skips the host log */
};

typedef struct canopy_batch_info {
    int64_t n_candidates;   /* rows that survive under the envelope of the trials */
    int64_t nnz_candidates;
    double candidate_ms;    /* device time: envelope pass + compaction */
    double trials_ms;       /* device time: all trials over the compact CSR + their final reduce */
    double host_log_ms;     /* host time: log(p) of n_trials * n_events values, staging, transpose */
} canopy_batch_info;

/* Enqueue the batch. device_out: NULL, or a device buffer of n_trials * CANOPY_PARTIAL_DOUBLES doubles that
   receives the records (e.g. this rank's slot of an all-gather buffer). Blocks on the host while it
   sizes the compact CSR; the trial kernels are left running on the context's stream. */
CANOPY_API canopy_status canopy_quantify_batch_async(canopy_ctx* ctx, canopy_model* model, const double* p_trials,
                                                     int32_t n_trials, const canopy_quant_params* params,
                                                     uint32_t batch_flags, void* device_out);
/* Wait, copy the n_trials records of this shard to the host. info may be NULL. */
CANOPY_API canopy_status canopy_quantify_batch_fetch(canopy_ctx* ctx, canopy_partial* partials, int32_t n_trials,
                                                     canopy_batch_info* info);
/* Synchronous, single shard: results[n_trials]. info may be NULL. */
CANOPY_API canopy_status canopy_quantify_batch(canopy_ctx* ctx, canopy_model* model, const double* p_trials,
                                               int32_t n_trials, const canopy_quant_params* params,
                                               uint32_t batch_flags, canopy_quant_result* results,
                                               canopy_batch_info* info);

/* ---- K6: direct Monte-Carlo propagation through the gate DAG ------------------------------------ */
/*
 * Reference vocabulary: Approximation::kMonteCarlo (src/mef/openpsa/settings.h:19), num_trials and
 * seed (settings.h:324-325), gates = Formula{connective, min, args} with complemented arguments
 * (src/mef/openpsa/event/formula.h:6-148), connective order of event.h:143-157. The reference has
 * no simulator; semantics here (oracle/canopy_oracle.c restates them bit for bit):
 *   sample n of basic event j is TRUE iff (mix64(ek_j + n) >> 11) < ceil(p_j * 2^53), a counter-based
 *   draw (ek_j derived from seed and j), so any sample range gives the same bits on any device;
 *   64 samples are packed into one u64 word per node; gates are evaluated with bitwise logic
 *   (ATLEAST through a bit-sliced adder); hits = popcount of the top node's words.
 * Integer result: bit-exact against the oracle.
 */
typedef enum canopy_connective { /* same order as the reference's Connective enum */
    CANOPY_GATE_AND = 0,
    CANOPY_GATE_OR = 1,
    CANOPY_GATE_ATLEAST = 2, /* K-of-N, K in gate_min */
    CANOPY_GATE_XOR = 3,
    CANOPY_GATE_NOT = 4,
    CANOPY_GATE_NAND = 5,
    CANOPY_GATE_NOR = 6,
    CANOPY_GATE_NULL = 7     /* single-argument pass-through */
} canopy_connective;

typedef struct canopy_dag_view {
    int32_t n_events;
    int32_t n_gates;
    const int32_t* gate_type; /* [n_gates] canopy_connective */
    const int32_t* gate_min;  /* [n_gates] K of ATLEAST gates (ignored otherwise); NULL if there are none */
    const int32_t* arg_ptr;   /* [n_gates+1] */
    const int32_t* args;      /* node ids: j < n_events is basic event j, n_events + g is gate g (g smaller
                                 than the gate that uses it: topological order); ~id = complemented argument */
    int32_t top;              /* node id whose probability is estimated */
    int32_t reserved;
} canopy_dag_view;

typedef struct canopy_dag canopy_dag;

CANOPY_API canopy_status canopy_dag_upload(canopy_ctx* ctx, const canopy_dag_view* dag, canopy_dag** out);
CANOPY_API void canopy_dag_free(canopy_dag* dag);
CANOPY_API canopy_status canopy_dag_set_probabilities(canopy_ctx* ctx, canopy_dag* dag, const double* p);

/* Settings::batch_size / sample_size (settings.h:129-151,326-327) exist in the reference as names with default 1 and no
   stated meaning; they are mirrored here as what a sampling loop of that shape computes: a run draws
   num_trials * batch_size * sample_size Bernoulli samples per basic event ("num_trials batches of batch_size trials of
   sample_size samples"), hits and num_trials of the result count samples. 0 is read as 1. The sample definition is
   counter-based, so a run may be cut into any number of calls by first_sample without changing a bit. */
typedef struct canopy_mc_params {
    uint64_t seed;        /* settings.h:324 */
    int64_t first_sample; /* global index of this shard's first sample, a multiple of 64 */
    int64_t num_trials;   /* settings.h:325 */
    int32_t batch_size;   /* settings.h:326, default 1 */
    int32_t sample_size;  /* settings.h:327, default 1 */
} canopy_mc_params;
CANOPY_API void canopy_mc_params_default(canopy_mc_params* params); /* seed 0, first_sample 0, num_trials 1000, 1, 1 */

typedef struct canopy_mc_result {
    int64_t hits;
    int64_t num_trials;
    double estimate;   /* hits / num_trials */
    double std_error;  /* sqrt(estimate * (1 - estimate) / num_trials) */
    double elapsed_kernel_ms;
} canopy_mc_result;

CANOPY_API canopy_status canopy_mc_run_async(canopy_ctx* ctx, const canopy_dag* dag, const canopy_mc_params* params);
CANOPY_API canopy_status canopy_mc_fetch(canopy_ctx* ctx, canopy_mc_result* result);
CANOPY_API canopy_status canopy_mc_run(canopy_ctx* ctx, const canopy_dag* dag, const canopy_mc_params* params,
                                       canopy_mc_result* result);

/* Global {hits, trials} of the last canopy_mc_run_async on a context with a communicator (see canopy_comm_*). */
CANOPY_API canopy_status canopy_comm_mc_fetch(canopy_comm* comm, canopy_mc_result* result);

/* ---- K0: fault tree -> cut sets / gate DAG (host only, no device needed) --------------------------- */
/*
 * A small reader for the Open-PSA Model Exchange Format subset that fault trees need
 * (<define-fault-tree>, <define-gate> with and / or / atleast min=K / not / xor / nand / nor / a bare
 * event reference, <define-basic-event> with <float value=.../>, <define-house-event> with
 * <constant value="true|false"/>, <model-data>), and the lowering the kernels consume. It stands in for
 * the reference's src/io/xml + src/mef/openpsa Initializer on this path (they do not build in this
 * toolchain) and is restated from the format, not from their text:
 *   - house events are folded as Boolean constants (reference event.h:34-57),
 *   - the top event is the gate no other gate refers to (reference fault_tree.h:151-186),
 *   - p[j] = the basic event's constant expression (reference event.h:93-96, expr/constant.h:23-31).
 * canopy_ft_cut_sets: MOCUS-style top-down expansion with minimisation (coherent trees: and / or /
 * atleast; a tree with negation returns CANOPY_ERR_UNSUPPORTED and must use the Monte-Carlo DAG),
 * rows = minimal cut sets with ascending basic-event indices = the CSR of canopy_model_upload.
 */
typedef struct canopy_ft canopy_ft;

typedef struct canopy_ft_info {
    int32_t n_basic_events;
    int32_t n_gates;
    int32_t n_house_events;
    int32_t top_gate;   /* index into the gate list */
    int32_t n_dag_args; /* total arguments of the folded gate DAG (size of `args` in canopy_ft_dag) */
    int32_t reserved;
} canopy_ft_info;

/* err (may be NULL) receives a NUL-terminated message on failure. */
CANOPY_API canopy_status canopy_ft_parse(const char* xml_text, size_t len, canopy_ft** out, char* err, size_t err_len);
CANOPY_API canopy_status canopy_ft_parse_file(const char* path, canopy_ft** out, char* err, size_t err_len);
CANOPY_API void canopy_ft_free(canopy_ft* ft);
CANOPY_API canopy_status canopy_ft_get_info(const canopy_ft* ft, canopy_ft_info* info);
CANOPY_API const char* canopy_ft_basic_event_name(const canopy_ft* ft, int32_t j);
CANOPY_API const char* canopy_ft_gate_name(const canopy_ft* ft, int32_t g);
CANOPY_API canopy_status canopy_ft_probabilities(const canopy_ft* ft, double* p /* [n_basic_events] */);
/* Gate DAG in canopy_dag_view layout (house events folded, gates in topological order); arrays sized
   from canopy_ft_info: gate_type/gate_min [n_gates], arg_ptr [n_gates+1], args [n_dag_args]. */
CANOPY_API canopy_status canopy_ft_dag(const canopy_ft* ft, int32_t* gate_type, int32_t* gate_min, int32_t* arg_ptr,
                                       int32_t* args, int32_t* top);
/* Minimal cut sets up to limit_order (0 = no limit). First call with row_ptr = col_idx = NULL to get the
   sizes, then with arrays of n_sets+1 and nnz entries. */
CANOPY_API canopy_status canopy_ft_cut_sets(const canopy_ft* ft, int32_t limit_order, int64_t* n_sets, int64_t* nnz,
                                            int64_t* row_ptr, int32_t* col_idx);

/* ---- per-launch timing over a region (HIP events on the context's stream) -------------------- */
/*
 * Between begin and end every canopy_quantify_async records an event triple around its row
 * kernel and its final reduce (up to max_launches launches; later ones are not timed).
 * end synchronises the stream and returns the summed device times and the launch count.
 */
CANOPY_API canopy_status canopy_timing_begin(canopy_ctx* ctx, int32_t max_launches);
CANOPY_API canopy_status canopy_timing_end(canopy_ctx* ctx, double* row_kernel_ms_sum,
                                           double* total_ms_sum, int32_t* n_launches);

/* Launch i (0-based) of the last closed timing region: device ms of its row kernel(s), and of the whole pass
   (final reduce and, with a communicator, the all-reduce included). */
CANOPY_API canopy_status canopy_timing_get(canopy_ctx* ctx, int32_t i, double* row_kernel_ms, double* total_ms);

/* ---- synthetic inputs (host logic) ----------------------------------------------------------- */
/* p_j = 10^(-8 + 5 u_j), u_j uniform [0,1) from the same counter-based generator: p[n_events]. */
CANOPY_API canopy_status canopy_synthetic_probabilities(uint64_t seed, int32_t n_events, double* p);

/* ---- sharding (host logic, no device needed) ------------------------------------------------ */
/*
 * Contiguous row ranges balanced by nnz: bounds[n_shards+1], bounds[0] = 0,
 * bounds[n_shards] = n_rows; shard k owns rows [bounds[k], bounds[k+1]).
 */
CANOPY_API canopy_status canopy_shard_rows(const int64_t* row_ptr, int64_t n_rows, int32_t n_shards,
                                           int64_t* bounds);
/* The same rule for rows [0, n_rows) of the synthetic family (canopy_model_synthetic) without materialising it on
   the host: the row lengths are generated and scanned on the context's device (no columns), the bounds searched
   there. Every rank of a strong-scaling run calls this and then generates only its own shard. */
CANOPY_API canopy_status canopy_synthetic_shard_bounds(canopy_ctx* ctx, uint64_t seed, int64_t n_rows, int32_t avg_nnz,
                                                       int32_t n_shards, int64_t* bounds);

#ifdef __cplusplus
}
#endif

#endif /* CANOPY_CANOPY_H */

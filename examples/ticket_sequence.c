/* ticket_sequence.c -- a plain C (C99) consumer of include/canopy.h: a sequence of quantifications, each with its own
 * probability vector, each result delivered to the host while later passes are in flight (per-pass tickets).
 * What an analysis loop over fresh BasicEvent::p() values (reference src/mef/openpsa/event/event.h:93-96) looks like
 * against the C ABI.
 *   gcc -std=c99 -O2 -Iinclude examples/ticket_sequence.c -Lcanopy_amd -lcanopy -Wl,-rpath,$PWD/canopy_amd -lm -o ticket_sequence
 *   ./ticket_sequence [rows=2000000] [events=50000] [vectors=12]
 * Prints one JSON object: the per-vector results (rare-event sum, MCUB, survivors) in enqueue order. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "canopy.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        canopy_status st_ = (call);                                                                   \
        if (st_ != CANOPY_OK) {                                                                       \
            fprintf(stderr, "%s: %s (%s)\n", #call, canopy_status_string(st_), ctx ? canopy_last_error(ctx) : ""); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

int main(int argc, char** argv) {
    const int64_t rows = argc > 1 ? atoll(argv[1]) : 2000000;
    const int32_t events = argc > 2 ? atoi(argv[2]) : 50000;
    const int n_vectors = argc > 3 ? atoi(argv[3]) : 12;
    const uint64_t seed = 20250704;
    canopy_ctx* ctx = NULL;
    canopy_model* model = NULL;
    canopy_ctx_params cp = {0, 0, NULL};
    canopy_quant_params prm;
    canopy_ticket inflight[3];
    int n_inflight = 0, k, printed = 0;
    double* base = (double*)malloc(sizeof(double) * (size_t)events);
    double* p = (double*)malloc(sizeof(double) * (size_t)events);
    if (!base || !p) return 2;
    CHECK(canopy_ctx_create(&cp, &ctx));
    CHECK(canopy_synthetic_probabilities(seed, events, base));
    CHECK(canopy_model_synthetic(ctx, seed, 0, rows, events, 6, &model));
    canopy_quant_params_default(&prm);
    printf("{\"rows\": %lld, \"events\": %d, \"results\": [", (long long)rows, events);
    for (k = 0; k <= n_vectors + 2; ++k) {
        if (k < n_vectors) { /* vector k: every probability scaled by a factor between 0.5 and 1.5 */
            const double f = 0.5 + (double)k / (double)(n_vectors > 1 ? n_vectors - 1 : 1);
            int32_t j;
            for (j = 0; j < events; ++j) p[j] = fmin(1.0, base[j] * f);
            CHECK(canopy_model_set_probabilities(ctx, model, p)); /* asynchronous: nothing in flight is disturbed */
            CHECK(canopy_quantify_enqueue(ctx, model, &prm, &inflight[n_inflight++]));
        }
        if (n_inflight == 3 || (k >= n_vectors && n_inflight > 0)) { /* the oldest pass: wait for IT only */
            canopy_partial rec;
            canopy_quant_result r;
            CHECK(canopy_ticket_fetch(ctx, inflight[0], &rec));
            CHECK(canopy_combine_partials(&rec, 1, 0, &r));
            printf("%s{\"rare_event\": %.17g, \"mcub\": %.17g, \"n_kept\": %lld}", printed++ ? ", " : "", r.rare_event, r.mcub,
                   (long long)r.n_kept);
            inflight[0] = inflight[1];
            inflight[1] = inflight[2];
            --n_inflight;
        }
    }
    printf("]}\n");
    canopy_model_free(model);
    canopy_ctx_free(ctx);
    free(base);
    free(p);
    return printed == n_vectors ? 0 : 3;
}

/*
 * canopy-cpp.h -- C++ RAII wrappers over the C ABI of canopy.h.
 *
 * The reference ships this header with nothing but the C++ guard, <memory> and the include of
 * canopy.h (reference include/canopy-cpp.h:34-40): the shape of a unique_ptr + deleter wrapper
 * header. This fills it in for the quantification path. Header-only; no exceptions are thrown by
 * the library, these wrappers turn a failing status into canopy::error.
 */
#pragma once

#ifndef __cplusplus
#error "This header is for C++ only"
#endif

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "canopy.h"

namespace canopy {

struct error : std::runtime_error {
    canopy_status status;
    error(canopy_status s, const std::string& where, const char* detail = nullptr)
        : std::runtime_error(where + ": " + canopy_status_string(s) + (detail && *detail ? std::string(": ") + detail : "")),
          status(s) {}
};

struct ctx_deleter { void operator()(canopy_ctx* p) const { canopy_ctx_free(p); } };
struct model_deleter { void operator()(canopy_model* p) const { canopy_model_free(p); } };
using ctx_ptr = std::unique_ptr<canopy_ctx, ctx_deleter>;
using model_ptr = std::unique_ptr<canopy_model, model_deleter>;

/* settings.h:314-332 defaults */
struct quant_params : canopy_quant_params {
    quant_params() { canopy_quant_params_default(this); }
};

class context {
public:
    explicit context(int device = 0, void* stream = nullptr, int n_blocks = 0) {
        canopy_ctx_params p{device, n_blocks, stream};
        canopy_ctx* raw = nullptr;
        canopy_status s = canopy_ctx_create(&p, &raw);
        if (s != CANOPY_OK) throw error(s, "canopy_ctx_create");
        ctx_.reset(raw);
    }
    canopy_ctx* get() const { return ctx_.get(); }

    model_ptr upload(int64_t n_rows, int32_t n_events, const int64_t* row_ptr, const int32_t* col_idx,
                     int64_t row_offset = 0, uint32_t flags = CANOPY_UPLOAD_VALIDATE) const {
        canopy_csr_view v{n_rows, n_events, 0, row_ptr, col_idx, row_offset};
        canopy_model* m = nullptr;
        check(canopy_model_upload(get(), &v, flags, &m), "canopy_model_upload");
        return model_ptr(m);
    }
    model_ptr synthetic(uint64_t seed, int64_t first_row, int64_t n_rows, int32_t n_events, int32_t avg_nnz) const {
        canopy_model* m = nullptr;
        check(canopy_model_synthetic(get(), seed, first_row, n_rows, n_events, avg_nnz, &m), "canopy_model_synthetic");
        return model_ptr(m);
    }
    void set_probabilities(const model_ptr& m, const std::vector<double>& p) const {
        if (p.size() != (size_t)model_info(m).n_events)
            throw error(CANOPY_ERR_INVALID_ARGUMENT, "set_probabilities", "p must hold n_events probabilities");
        check(canopy_model_set_probabilities(get(), m.get(), p.data()), "canopy_model_set_probabilities");
    }
    canopy_quant_result quantify(const model_ptr& m, const canopy_quant_params& params = quant_params()) const {
        canopy_quant_result r;
        check(canopy_quantify(get(), m.get(), &params, &r), "canopy_quantify");
        return r;
    }
    void quantify_async(const model_ptr& m, const canopy_quant_params& params = quant_params()) const {
        check(canopy_quantify_async(get(), m.get(), &params), "canopy_quantify_async");
    }
    canopy_partial fetch() const {
        canopy_partial p;
        check(canopy_quantify_fetch(get(), &p), "canopy_quantify_fetch");
        return p;
    }
    // one result per pass while later passes run: enqueue() hands out a ticket, fetch(ticket) waits for that pass only
    // (a ticket stays fetchable until 64 later passes have been enqueued on the context)
    canopy_ticket enqueue(const model_ptr& m, const canopy_quant_params& params = quant_params()) const {
        canopy_ticket t = 0;
        check(canopy_quantify_enqueue(get(), m.get(), &params, &t), "canopy_quantify_enqueue");
        return t;
    }
    canopy_partial fetch(canopy_ticket t) const {
        canopy_partial p;
        check(canopy_ticket_fetch(get(), t, &p), "canopy_ticket_fetch");
        return p;
    }
    void set_log_probabilities(const model_ptr& m, const std::vector<double>& logp) const {
        if (logp.size() != (size_t)model_info(m).n_events)
            throw error(CANOPY_ERR_INVALID_ARGUMENT, "set_log_probabilities", "logp must hold n_events values");
        check(canopy_model_set_log_probabilities(get(), m.get(), logp.data()), "canopy_model_set_log_probabilities");
    }
    // K probability vectors against one model (p_trials: n_trials x n_events, row-major); results[k] is what
    // quantify() returns after set_probabilities(p_trials[k]). info may be null.
    std::vector<canopy_quant_result> quantify_batch(const model_ptr& m, const std::vector<double>& p_trials, int32_t n_trials,
                                                    const canopy_quant_params& params = quant_params(),
                                                    canopy_batch_info* info = nullptr) const {
        if (n_trials < 0 || p_trials.size() != (size_t)n_trials * (size_t)model_info(m).n_events)
            throw error(CANOPY_ERR_INVALID_ARGUMENT, "quantify_batch", "p_trials must hold n_trials x n_events values");
        std::vector<canopy_quant_result> r((size_t)n_trials);
        check(canopy_quantify_batch(get(), m.get(), p_trials.data(), n_trials, &params, 0, r.data(), info), "canopy_quantify_batch");
        return r;
    }
    canopy_model_info model_info(const model_ptr& m) const {
        canopy_model_info i;
        check(canopy_model_get_info(m.get(), &i), "canopy_model_get_info");
        return i;
    }
    void synchronize() const { check(canopy_ctx_synchronize(get()), "canopy_ctx_synchronize"); }

private:
    void check(canopy_status s, const char* where) const {
        if (s != CANOPY_OK) throw error(s, where, canopy_last_error(get()));
    }
    ctx_ptr ctx_;
};

} // namespace canopy

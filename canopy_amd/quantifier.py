"""Host-side mirror of the C ABI (include/canopy.h) for the cut-set quantification path.

Thin RAII objects over the opaque handles -- the Python analogue of include/canopy-cpp.h, which
follows the reference's unique_ptr-wrapper header shape (reference include/canopy-cpp.h:34-40).
Names, argument meaning and defaults are the ABI's, which in turn mirror
mef::openpsa::Settings (reference src/mef/openpsa/settings.h:19,314-332).

Everything here goes through libcanopy.so; nothing falls back to numpy or to oracle/.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _capi as capi


@dataclass
class QuantResult:
    rare_event: float
    mcub: float
    n_kept: int
    argmax_row: int
    max_logp: float
    elapsed_kernel_ms: float = -1.0


def _as(arr, dtype):
    a = np.ascontiguousarray(arr, dtype=dtype)
    return a


class Context:
    """One device, one stream (canopy_ctx). `stream` is a raw hipStream_t integer or None."""

    def __init__(self, device=0, stream=None, n_blocks=0):
        self._h = C.c_void_p()
        p = capi.CtxParams(device=device, n_blocks=n_blocks, stream=stream)
        capi.check(capi.lib().canopy_ctx_create(C.byref(p), C.byref(self._h)), "canopy_ctx_create")
        self.device = device

    def close(self):
        if self._h:
            capi.lib().canopy_ctx_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        capi.check(capi.lib().canopy_ctx_synchronize(self._h), "canopy_ctx_synchronize", self._h)

    def set_pipeline(self, enable):
        """Overlap the tail of a sum-only pass with the next pass's row kernel (canopy_ctx_set_pipeline; default on)."""
        capi.check(capi.lib().canopy_ctx_set_pipeline(self._h, int(enable)), "canopy_ctx_set_pipeline", self._h)

    def partials_device_ptr(self):
        return capi.lib().canopy_ctx_partials_device(self._h)

    # -- models ---------------------------------------------------------------------------------
    def upload(self, row_ptr, col_idx, n_events, row_offset=0, validate=True):
        row_ptr = _as(row_ptr, np.int64)
        col_idx = _as(col_idx, np.int32)
        view = capi.CsrView(n_rows=len(row_ptr) - 1, n_events=n_events, reserved=0,
                            row_ptr=row_ptr.ctypes.data, col_idx=col_idx.ctypes.data if col_idx.size else None,
                            row_offset=row_offset)
        h = C.c_void_p()
        capi.check(capi.lib().canopy_model_upload(self._h, C.byref(view), capi.UPLOAD_VALIDATE if validate else 0,
                                                  C.byref(h)), "canopy_model_upload", self._h)
        return Model(self, h)

    def synthetic(self, seed, n_rows, n_events, avg_nnz, first_row=0):
        h = C.c_void_p()
        capi.check(capi.lib().canopy_model_synthetic(self._h, seed, first_row, n_rows, n_events, avg_nnz,
                                                     C.byref(h)), "canopy_model_synthetic", self._h)
        return Model(self, h)

    # -- quantification ---------------------------------------------------------------------------
    def quantify_async(self, model, params=None):
        capi.check(capi.lib().canopy_quantify_async(self._h, model.handle, C.byref(params) if params else None),
                   "canopy_quantify_async", self._h)

    def enqueue(self, model, params=None):
        """quantify_async + a ticket: the pass's record can be fetched with ticket_fetch() while later passes run
        (canopy_quantify_enqueue). A ticket stays fetchable until 64 later passes have been enqueued on the context."""
        t = C.c_uint64()
        capi.check(capi.lib().canopy_quantify_enqueue(self._h, model.handle, C.byref(params) if params else None, C.byref(t)),
                   "canopy_quantify_enqueue", self._h)
        return t.value

    def ticket_fetch(self, ticket):
        """The shard record of the pass `ticket` (waits for that pass's lane only)."""
        part = capi.Partial()
        capi.check(capi.lib().canopy_ticket_fetch(self._h, ticket, C.byref(part)), "canopy_ticket_fetch", self._h)
        return part

    def ticket_done(self, ticket):
        d = C.c_int32()
        capi.check(capi.lib().canopy_ticket_done(self._h, ticket, C.byref(d)), "canopy_ticket_done", self._h)
        return bool(d.value)

    def fetch_partial(self):
        part = capi.Partial()
        capi.check(capi.lib().canopy_quantify_fetch(self._h, C.byref(part)), "canopy_quantify_fetch", self._h)
        return part

    def quantify(self, model, params=None):
        res = capi.QuantResult()
        capi.check(capi.lib().canopy_quantify(self._h, model.handle, C.byref(params) if params else None,
                                              C.byref(res)), "canopy_quantify", self._h)
        return QuantResult(res.rare_event, res.mcub, res.n_kept, res.argmax_row, res.max_logp,
                           res.elapsed_kernel_ms)

    # -- captured passes (hipGraph) ----------------------------------------------------------------
    def graph_capture(self, model, params=None):
        g = C.c_void_p()
        capi.check(capi.lib().canopy_graph_capture(self._h, model.handle, C.byref(params) if params else None, C.byref(g)),
                   "canopy_graph_capture", self._h)
        return g

    def graph_launch(self, graph):
        capi.check(capi.lib().canopy_graph_launch(self._h, graph), "canopy_graph_launch", self._h)

    @staticmethod
    def graph_free(graph):
        capi.lib().canopy_graph_free(graph)

    def fetch_importance(self, model):
        """contrib[j] = sum of P_i over kept cut sets containing event j (last pass with QUANT_IMPORTANCE)."""
        out = np.empty(model.info().n_events, dtype=np.float64)
        capi.check(capi.lib().canopy_fetch_importance(self._h, model.handle, out.ctypes.data), "canopy_fetch_importance",
                   self._h)
        return out

    def set_importance_buffer(self, device_ptr, n_events):
        capi.check(capi.lib().canopy_ctx_set_importance_buffer(self._h, device_ptr, n_events),
                   "canopy_ctx_set_importance_buffer", self._h)

    # -- batched quantification: K probability vectors against one model (N4) ----------------------
    def quantify_batch_async(self, model, p_trials, params=None, log_input=False, device_out=None):
        """p_trials: [n_trials, n_events]. device_out: raw device pointer (int) for the records, or None."""
        p_trials = _as(p_trials, np.float64)
        if p_trials.ndim != 2 or p_trials.shape[1] != model.info().n_events:
            raise ValueError("p_trials must be [n_trials, n_events]")
        capi.check(capi.lib().canopy_quantify_batch_async(self._h, model.handle, p_trials.ctypes.data, p_trials.shape[0],
                                                          C.byref(params) if params else None,
                                                          capi.BATCH_LOG_INPUT if log_input else 0, device_out),
                   "canopy_quantify_batch_async", self._h)
        return p_trials.shape[0]

    def fetch_batch_partials(self, n_trials):
        parts = (capi.Partial * n_trials)()
        info = capi.BatchInfo()
        capi.check(capi.lib().canopy_quantify_batch_fetch(self._h, parts, n_trials, C.byref(info)),
                   "canopy_quantify_batch_fetch", self._h)
        return list(parts), info

    def quantify_batch(self, model, p_trials, params=None, log_input=False):
        """-> (list of QuantResult, BatchInfo); trial k == quantify() after set_probabilities(p_trials[k])."""
        n = self.quantify_batch_async(model, p_trials, params, log_input)
        parts, info = self.fetch_batch_partials(n)
        flags = params.flags if params else 0
        return [combine_partials([q], flags) for q in parts], info

    def last_kernel_ms(self):
        a, b = C.c_double(), C.c_double()
        capi.check(capi.lib().canopy_last_kernel_ms(self._h, C.byref(a), C.byref(b)), "canopy_last_kernel_ms", self._h)
        return a.value, b.value

    def last_kernel_variant(self):
        """The CANOPY_KERNEL_* id the last pass ran (what AUTO resolved to)."""
        return capi.lib().canopy_last_kernel_variant(self._h)

    def timing_begin(self, max_launches):
        capi.check(capi.lib().canopy_timing_begin(self._h, max_launches), "canopy_timing_begin", self._h)

    def timing_end(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int32()
        capi.check(capi.lib().canopy_timing_end(self._h, C.byref(a), C.byref(b), C.byref(n)), "canopy_timing_end", self._h)
        return a.value, b.value, n.value

    def timing_launches(self, n):
        """[(row_kernel_ms, pass_ms)] of the first n launches of the last closed timing region."""
        out = []
        a, b = C.c_double(), C.c_double()
        for i in range(n):
            capi.check(capi.lib().canopy_timing_get(self._h, i, C.byref(a), C.byref(b)), "canopy_timing_get", self._h)
            out.append((a.value, b.value))
        return out

    def last_filter_candidates(self):
        """(rows flagged by the last fetched FILTER_REFINE pass, rows it looked at) or None."""
        a, b = C.c_int64(), C.c_int64()
        if capi.lib().canopy_last_filter_candidates(self._h, C.byref(a), C.byref(b)) != capi.OK:
            return None
        return a.value, b.value

    def fetch_row_logp(self, model, first_row=0, n=None):
        n = model.info().n_rows - first_row if n is None else n
        out = np.empty(n, dtype=np.float64)
        capi.check(capi.lib().canopy_fetch_row_logp(self._h, model.handle, first_row, n, out.ctypes.data),
                   "canopy_fetch_row_logp", self._h)
        return out

    def fetch_keep_mask(self, model, first_word=0, n_words=None):
        total = (model.info().n_rows + 63) // 64
        n_words = total - first_word if n_words is None else n_words
        out = np.empty(n_words, dtype=np.uint64)
        capi.check(capi.lib().canopy_fetch_keep_mask(self._h, model.handle, first_word, n_words, out.ctypes.data),
                   "canopy_fetch_keep_mask", self._h)
        return out


class Comm:
    """K5 inside the library (canopy_comm): the single RCCL all-reduce is issued by libcanopy.so itself, right
    behind the final-reduce kernel of every pass on the context. One process per GPU: rank 0 makes the id
    (Comm.unique_id()), the launcher hands its 128 bytes to every rank, every rank constructs Comm(ctx, id, world,
    rank) -- a collective call. Free it before its context."""

    def __init__(self, ctx, comm_id, world_size, rank):
        cid = capi.CommId()
        C.memmove(C.byref(cid), bytes(comm_id), 128)
        self._ctx = ctx
        self._h = C.c_void_p()
        capi.check(capi.lib().canopy_comm_init_rank(ctx.handle, C.byref(cid), world_size, rank, C.byref(self._h)),
                   "canopy_comm_init_rank", ctx.handle)
        self.world, self.rank = world_size, rank

    @staticmethod
    def unique_id():
        cid = capi.CommId()
        capi.check(capi.lib().canopy_comm_get_unique_id(C.byref(cid)), "canopy_comm_get_unique_id")
        return bytes(bytearray(C.string_at(C.byref(cid), 128)))

    @property
    def handle(self):
        return self._h

    def rccl_version(self):
        v = C.c_int32()
        capi.check(capi.lib().canopy_comm_info(self._h, None, None, C.byref(v)), "canopy_comm_info")
        return v.value

    def fetch(self, flags=0):
        """Combined result of the last pass (waits for the stream; identical on every rank)."""
        res = capi.QuantResult()
        capi.check(capi.lib().canopy_comm_fetch(self._h, flags, C.byref(res), None), "canopy_comm_fetch", self._ctx.handle)
        return QuantResult(res.rare_event, res.mcub, res.n_kept, res.argmax_row, res.max_logp, -1.0)

    def ticket_fetch(self, ticket, flags=0):
        """Combined result of the pass `ticket` (canopy_comm_ticket_fetch: waits for that pass's lane only)."""
        res = capi.QuantResult()
        capi.check(capi.lib().canopy_comm_ticket_fetch(self._h, ticket, flags, C.byref(res), None), "canopy_comm_ticket_fetch",
                   self._ctx.handle)
        return QuantResult(res.rare_event, res.mcub, res.n_kept, res.argmax_row, res.max_logp, -1.0)

    def ticket_partials(self, ticket):
        parts = (capi.Partial * self.world)()
        capi.check(capi.lib().canopy_comm_ticket_fetch(self._h, ticket, 0, None, parts), "canopy_comm_ticket_fetch",
                   self._ctx.handle)
        return list(parts)

    def set_mode(self, mode):
        """capi.COMM_CHAINED (default) / COMM_UNORDERED / COMM_OWN_STREAM: how consecutive collectives are ordered."""
        capi.check(capi.lib().canopy_comm_set_mode(self._h, mode), "canopy_comm_set_mode", self._ctx.handle)

    def partials(self):
        parts = (capi.Partial * self.world)()
        capi.check(capi.lib().canopy_comm_fetch(self._h, 0, None, parts), "canopy_comm_fetch", self._ctx.handle)
        return list(parts)

    def last_collective_ms(self):
        ms = C.c_double()
        capi.check(capi.lib().canopy_comm_last_collective_ms(self._h, C.byref(ms)), "canopy_comm_last_collective_ms",
                   self._ctx.handle)
        return ms.value

    def mc_fetch(self):
        res = capi.McResult()
        capi.check(capi.lib().canopy_comm_mc_fetch(self._h, C.byref(res)), "canopy_comm_mc_fetch", self._ctx.handle)
        return res

    def close(self):
        if self._h:
            capi.lib().canopy_comm_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Dag:
    """A gate DAG resident on the device (canopy_dag) for the Monte-Carlo path (K6)."""

    def __init__(self, ctx, dag):
        self._ctx = ctx
        self._keep = [_as(dag["gate_type"], np.int32), _as(dag["gate_min"], np.int32), _as(dag["arg_ptr"], np.int32),
                      _as(dag["args"], np.int32)]
        gt, gm, ap, ar = self._keep
        view = capi.DagView(n_events=dag["n_events"], n_gates=len(gt), gate_type=gt.ctypes.data, gate_min=gm.ctypes.data,
                            arg_ptr=ap.ctypes.data, args=ar.ctypes.data if ar.size else None, top=dag["top"], reserved=0)
        self._h = C.c_void_p()
        capi.check(capi.lib().canopy_dag_upload(ctx.handle, C.byref(view), C.byref(self._h)), "canopy_dag_upload", ctx.handle)
        self.n_events = dag["n_events"]

    def close(self):
        if self._h:
            capi.lib().canopy_dag_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_probabilities(self, p):
        p = _as(p, np.float64)
        if p.shape != (self.n_events,):
            raise ValueError("p must have n_events entries")
        capi.check(capi.lib().canopy_dag_set_probabilities(self._ctx.handle, self._h, p.ctypes.data),
                   "canopy_dag_set_probabilities", self._ctx.handle)

    def run_async(self, seed, num_trials, first_sample=0, batch_size=1, sample_size=1):
        prm = capi.McParams(seed=seed, first_sample=first_sample, num_trials=num_trials, batch_size=batch_size,
                            sample_size=sample_size)
        capi.check(capi.lib().canopy_mc_run_async(self._ctx.handle, self._h, C.byref(prm)), "canopy_mc_run_async",
                   self._ctx.handle)

    def fetch(self):
        res = capi.McResult()
        capi.check(capi.lib().canopy_mc_fetch(self._ctx.handle, C.byref(res)), "canopy_mc_fetch", self._ctx.handle)
        return res

    def run(self, seed, num_trials, first_sample=0, batch_size=1, sample_size=1):
        self.run_async(seed, num_trials, first_sample, batch_size, sample_size)
        return self.fetch()


class Model:
    """A CSR cut-set shard resident in HBM (canopy_model)."""

    def __init__(self, ctx, handle):
        self._ctx = ctx
        self._h = handle

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            capi.lib().canopy_model_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        info = capi.ModelInfo()
        capi.check(capi.lib().canopy_model_get_info(self._h, C.byref(info)), "canopy_model_get_info")
        return info

    def download(self):
        info = self.info()
        row_ptr = np.empty(info.n_rows + 1, dtype=np.int64)
        col_idx = np.empty(info.nnz, dtype=np.int32)
        capi.check(capi.lib().canopy_model_download(self._ctx.handle, self._h, row_ptr.ctypes.data,
                                                    col_idx.ctypes.data if info.nnz else None),
                   "canopy_model_download", self._ctx.handle)
        return row_ptr, col_idx

    def prepare(self, kernel=capi.KERNEL_AUTO):
        capi.check(capi.lib().canopy_model_prepare(self._ctx.handle, self._h, kernel), "canopy_model_prepare",
                   self._ctx.handle)

    def set_probabilities(self, p):
        p = _as(p, np.float64)
        if p.shape != (self.info().n_events,):
            raise ValueError("p must have n_events entries")
        capi.check(capi.lib().canopy_model_set_probabilities(self._ctx.handle, self._h, p.ctypes.data),
                   "canopy_model_set_probabilities", self._ctx.handle)

    def set_log_probabilities(self, logp):
        """Caller-supplied natural logarithms (<= 0): no host log (canopy_model_set_log_probabilities)."""
        logp = _as(logp, np.float64)
        if logp.shape != (self.info().n_events,):
            raise ValueError("logp must have n_events entries")
        capi.check(capi.lib().canopy_model_set_log_probabilities(self._ctx.handle, self._h, logp.ctypes.data),
                   "canopy_model_set_log_probabilities", self._ctx.handle)


def default_params(cut_off=None, limit_order=None, flags=0, kernel=capi.KERNEL_AUTO):
    p = capi.QuantParams()
    capi.lib().canopy_quant_params_default(C.byref(p))
    if cut_off is not None:
        p.cut_off = cut_off
    if limit_order is not None:
        p.limit_order = limit_order
    p.flags = flags
    p.kernel = kernel
    return p


def importance_measures(contrib, p, rare_event):
    """Host arithmetic on the contributions: Fussell-Vesely, rare-event Birnbaum, risk reduction worth."""
    contrib = np.asarray(contrib, dtype=np.float64)
    p = np.asarray(p, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        fv = contrib / rare_event if rare_event > 0 else np.zeros_like(contrib)
        birnbaum = np.where(p > 0, contrib / p, 0.0)
        rrw = np.where(fv < 1, 1.0 / (1.0 - fv), np.inf)
    return {"fussell_vesely": fv, "birnbaum": birnbaum, "risk_reduction_worth": rrw}


def synthetic_probabilities(seed, n_events):
    p = np.empty(n_events, dtype=np.float64)
    capi.check(capi.lib().canopy_synthetic_probabilities(seed, n_events, p.ctypes.data),
               "canopy_synthetic_probabilities")
    return p


def shard_rows(row_ptr, n_shards):
    row_ptr = _as(row_ptr, np.int64)
    bounds = np.empty(n_shards + 1, dtype=np.int64)
    capi.check(capi.lib().canopy_shard_rows(row_ptr.ctypes.data, len(row_ptr) - 1, n_shards, bounds.ctypes.data),
               "canopy_shard_rows")
    return bounds


def synthetic_shard_bounds(ctx, seed, n_rows, avg_nnz, n_shards):
    """nnz-balanced contiguous row ranges of the synthetic family (canopy_shard_rows' rule, evaluated on the device)."""
    bounds = np.empty(n_shards + 1, dtype=np.int64)
    capi.check(capi.lib().canopy_synthetic_shard_bounds(ctx.handle, seed, n_rows, avg_nnz, n_shards, bounds.ctypes.data),
               "canopy_synthetic_shard_bounds", ctx.handle)
    return bounds


def combine_partials(parts, flags=0):
    """parts: sequence of capi.Partial (rank order) -> QuantResult (canopy_combine_partials)."""
    arr = (capi.Partial * len(parts))(*parts)
    res = capi.QuantResult()
    capi.check(capi.lib().canopy_combine_partials(arr, len(parts), flags, C.byref(res)), "canopy_combine_partials")
    return QuantResult(res.rare_event, res.mcub, res.n_kept, res.argmax_row, res.max_logp, -1.0)

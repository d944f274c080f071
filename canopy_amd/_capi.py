"""ctypes binding of include/canopy.h -- the same stub a maintainer of the reference would add
(INTEGRATION.md). No torch types cross this boundary: plain pointers and sizes only.

The library is loaded from the in-tree canopy_amd/libcanopy.so. There is no fallback: if the
HIP extension is missing this raises, and if no GPU is visible canopy_ctx_create returns
CANOPY_ERR_NO_DEVICE, which `check` turns into CanopyError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CANOPY_AMD_LIB") or os.path.join(_HERE, "libcanopy.so")  # (the override: A/B of two builds on one box)

PARTIAL_DOUBLES = 8

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_OUT_OF_MEMORY = -2
ERR_HIP = -3
ERR_NO_DEVICE = -4
ERR_UNSUPPORTED = -5
ERR_INVALID_MODEL = -6
ERR_STATE = -7
ERR_INTERNAL = -8
ERR_RCCL = -9

UPLOAD_VALIDATE = 1
UPLOAD_DEVICE_POINTERS = 2

APPROX_RARE_EVENT, APPROX_MCUB, APPROX_BOTH = 1, 2, 3
QUANT_CLAMP, QUANT_ROW_LOGP, QUANT_KEEP_MASK, QUANT_IMPORTANCE = 1, 2, 4, 8
BATCH_LOG_INPUT = 1
GATE_AND, GATE_OR, GATE_ATLEAST, GATE_XOR, GATE_NOT, GATE_NAND, GATE_NOR, GATE_NULL = range(8)
COMM_CHAINED, COMM_UNORDERED, COMM_OWN_STREAM = 0, 1, 2
KERNEL_AUTO, KERNEL_ROWS, KERNEL_ROWS_LDS, KERNEL_WAVE_STREAM, KERNEL_VALUE_SCATTER, KERNEL_FILTER_REFINE, KERNEL_FIXED_SUM, KERNEL_BAND_STREAM = range(8)


class CtxParams(C.Structure):
    _fields_ = [("device", C.c_int32), ("n_blocks", C.c_int32), ("stream", C.c_void_p)]


class CsrView(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("n_events", C.c_int32), ("reserved", C.c_int32),
                ("row_ptr", C.c_void_p), ("col_idx", C.c_void_p), ("row_offset", C.c_int64)]


class ModelInfo(C.Structure):
    _fields_ = [("n_rows", C.c_int64), ("nnz", C.c_int64), ("row_offset", C.c_int64),
                ("n_events", C.c_int32), ("max_row_len", C.c_int32),
                ("device_bytes", C.c_int64), ("stream_bytes", C.c_int64), ("algorithmic_bytes", C.c_int64),
                ("stream_bytes_wide", C.c_int64), ("stream_bytes_exact", C.c_int64), ("stream_bytes_rows", C.c_int64)]


class QuantParams(C.Structure):
    _fields_ = [("approximation", C.c_int32), ("limit_order", C.c_int32), ("cut_off", C.c_double),
                ("flags", C.c_uint32), ("kernel", C.c_int32)]


class Partial(C.Structure):
    _fields_ = [("rare_hi", C.c_double), ("rare_lo", C.c_double), ("log1p_hi", C.c_double),
                ("log1p_lo", C.c_double), ("n_kept", C.c_double), ("n_certain", C.c_double),
                ("max_logp", C.c_double), ("argmax_row", C.c_double)]


class BatchInfo(C.Structure):
    _fields_ = [("n_candidates", C.c_int64), ("nnz_candidates", C.c_int64), ("candidate_ms", C.c_double),
                ("trials_ms", C.c_double), ("host_log_ms", C.c_double)]


class DagView(C.Structure):
    _fields_ = [("n_events", C.c_int32), ("n_gates", C.c_int32), ("gate_type", C.c_void_p), ("gate_min", C.c_void_p),
                ("arg_ptr", C.c_void_p), ("args", C.c_void_p), ("top", C.c_int32), ("reserved", C.c_int32)]


class McParams(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_sample", C.c_int64), ("num_trials", C.c_int64),
                ("batch_size", C.c_int32), ("sample_size", C.c_int32)]


class McResult(C.Structure):
    _fields_ = [("hits", C.c_int64), ("num_trials", C.c_int64), ("estimate", C.c_double), ("std_error", C.c_double),
                ("elapsed_kernel_ms", C.c_double)]


class FtInfo(C.Structure):
    _fields_ = [("n_basic_events", C.c_int32), ("n_gates", C.c_int32), ("n_house_events", C.c_int32),
                ("top_gate", C.c_int32), ("n_dag_args", C.c_int32), ("reserved", C.c_int32)]


class CommId(C.Structure):
    _fields_ = [("bytes", C.c_char * 128)]


class QuantResult(C.Structure):
    _fields_ = [("rare_event", C.c_double), ("mcub", C.c_double), ("n_kept", C.c_int64),
                ("argmax_row", C.c_int64), ("max_logp", C.c_double), ("elapsed_kernel_ms", C.c_double)]


# name -> (restype, argtypes): every symbol include/canopy.h declares
SIGNATURES = {
    "canopy_status_string": (C.c_char_p, [C.c_int]),
    "canopy_version": (None, [C.POINTER(C.c_int)] * 3),
    "canopy_ctx_create": (C.c_int, [C.POINTER(CtxParams), C.POINTER(C.c_void_p)]),
    "canopy_ctx_free": (None, [C.c_void_p]),
    "canopy_last_error": (C.c_char_p, [C.c_void_p]),
    "canopy_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "canopy_ctx_set_pipeline": (C.c_int, [C.c_void_p, C.c_int32]),
    "canopy_model_upload": (C.c_int, [C.c_void_p, C.POINTER(CsrView), C.c_uint32, C.POINTER(C.c_void_p)]),
    "canopy_model_synthetic": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                         C.POINTER(C.c_void_p)]),
    "canopy_model_free": (None, [C.c_void_p]),
    "canopy_model_get_info": (C.c_int, [C.c_void_p, C.POINTER(ModelInfo)]),
    "canopy_model_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "canopy_model_set_probabilities": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "canopy_model_set_log_probabilities": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "canopy_model_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "canopy_quant_params_default": (None, [C.POINTER(QuantParams)]),
    "canopy_quantify_async": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(QuantParams)]),
    "canopy_quantify_fetch": (C.c_int, [C.c_void_p, C.POINTER(Partial)]),
    "canopy_quantify_enqueue": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(QuantParams), C.POINTER(C.c_uint64)]),
    "canopy_ticket_fetch": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(Partial)]),
    "canopy_ticket_done": (C.c_int, [C.c_void_p, C.c_uint64, C.POINTER(C.c_int32)]),
    "canopy_quantify": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(QuantParams), C.POINTER(QuantResult)]),
    "canopy_ctx_partials_device": (C.c_void_p, [C.c_void_p]),
    "canopy_ctx_set_gather_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "canopy_comm_get_unique_id": (C.c_int, [C.POINTER(CommId)]),
    "canopy_comm_init_rank": (C.c_int, [C.c_void_p, C.POINTER(CommId), C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "canopy_comm_init_all": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]),
    "canopy_comm_free": (None, [C.c_void_p]),
    "canopy_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "canopy_comm_quantify_all_async": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32,
                                                 C.POINTER(QuantParams)]),
    "canopy_comm_fetch": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(QuantResult), C.POINTER(Partial)]),
    "canopy_comm_ticket_fetch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(QuantResult), C.POINTER(Partial)]),
    "canopy_comm_set_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "canopy_comm_last_collective_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "canopy_comm_mc_fetch": (C.c_int, [C.c_void_p, C.POINTER(McResult)]),
    "canopy_synthetic_shard_bounds": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_int32, C.c_void_p]),
    "canopy_combine_partials": (C.c_int, [C.POINTER(Partial), C.c_int32, C.c_uint32, C.POINTER(QuantResult)]),
    "canopy_fetch_row_logp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "canopy_fetch_keep_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "canopy_graph_capture": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(QuantParams), C.POINTER(C.c_void_p)]),
    "canopy_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "canopy_graph_free": (None, [C.c_void_p]),
    "canopy_fetch_importance": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "canopy_ctx_set_importance_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "canopy_quantify_batch_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(QuantParams), C.c_uint32,
                                              C.c_void_p]),
    "canopy_quantify_batch_fetch": (C.c_int, [C.c_void_p, C.POINTER(Partial), C.c_int32, C.POINTER(BatchInfo)]),
    "canopy_quantify_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(QuantParams), C.c_uint32,
                                        C.POINTER(QuantResult), C.POINTER(BatchInfo)]),
    "canopy_dag_upload": (C.c_int, [C.c_void_p, C.POINTER(DagView), C.POINTER(C.c_void_p)]),
    "canopy_dag_free": (None, [C.c_void_p]),
    "canopy_dag_set_probabilities": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "canopy_mc_params_default": (None, [C.POINTER(McParams)]),
    "canopy_mc_run_async": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(McParams)]),
    "canopy_mc_fetch": (C.c_int, [C.c_void_p, C.POINTER(McResult)]),
    "canopy_mc_run": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(McParams), C.POINTER(McResult)]),
    "canopy_ft_parse": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "canopy_ft_parse_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "canopy_ft_free": (None, [C.c_void_p]),
    "canopy_ft_get_info": (C.c_int, [C.c_void_p, C.POINTER(FtInfo)]),
    "canopy_ft_basic_event_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "canopy_ft_gate_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "canopy_ft_probabilities": (C.c_int, [C.c_void_p, C.c_void_p]),
    "canopy_ft_dag": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "canopy_ft_cut_sets": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p,
                                     C.c_void_p]),
    "canopy_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "canopy_last_kernel_variant": (C.c_int32, [C.c_void_p]),
    "canopy_timing_begin": (C.c_int, [C.c_void_p, C.c_int32]),
    "canopy_timing_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int32)]),
    "canopy_timing_get": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "canopy_last_filter_candidates": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "canopy_synthetic_probabilities": (C.c_int, [C.c_uint64, C.c_int32, C.c_void_p]),
    "canopy_shard_rows": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
}

_lib = None


class CanopyError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        msg = "%s failed: %s (%d)" % (where, lib().canopy_status_string(status).decode(), status)
        if detail:
            msg += ": " + detail
        super().__init__(msg)


def lib():
    """Load libcanopy.so (once). Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "canopy_amd/libcanopy.so is missing: build it with `python -m canopy_amd.build` "
                "(there is no CPU fallback for the quantification path)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, where, ctx=None):
    if status != OK:
        detail = ""
        if ctx:
            detail = lib().canopy_last_error(ctx).decode()
        raise CanopyError(status, where, detail)

"""ctypes wrapper of oracle/libcanopy_oracle.so. TEST INFRASTRUCTURE ONLY (see canopy_oracle.h):
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, nowhere else."""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcanopy_oracle.so")


class OracleResult(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("rare_hi", "rare_lo", "log1p_hi", "log1p_lo", "n_kept", "n_certain",
                                          "max_logp", "argmax_row", "rare_event", "mcub", "elapsed_s")]


_lib = None


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("canopy_oracle.c", "canopy_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src):
        subprocess.run(["make", "-C", _HERE, "-B", "-s"], check=True)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        l = C.CDLL(LIB_PATH)
        l.oracle_quantify.restype = C.c_int
        l.oracle_quantify.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_double,
                                      C.c_int32, C.c_int32, C.POINTER(OracleResult), C.c_void_p, C.c_void_p]
        l.oracle_quantify_product_domain.restype = C.c_int
        l.oracle_quantify_product_domain.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.POINTER(C.c_double), C.POINTER(C.c_double)]
        l.oracle_combine.restype = None
        l.oracle_combine.argtypes = [C.POINTER(OracleResult), C.c_int32, C.POINTER(OracleResult)]
        l.oracle_draw.restype = C.c_uint64
        l.oracle_draw.argtypes = [C.c_uint64] * 4
        l.oracle_synth_row_len.restype = C.c_int32
        l.oracle_synth_row_len.argtypes = [C.c_uint64, C.c_int64, C.c_int32]
        l.oracle_synth_nnz.restype = C.c_int64
        l.oracle_synth_nnz.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int32]
        l.oracle_synth_csr.restype = C.c_int
        l.oracle_synth_csr.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p]
        l.oracle_synth_probabilities.restype = None
        l.oracle_synth_probabilities.argtypes = [C.c_uint64, C.c_int32, C.c_void_p]
        l.oracle_shard_rows.restype = None
        l.oracle_shard_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
        l.oracle_mc.restype = C.c_int64
        l.oracle_mc.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                C.c_uint64, C.c_int64, C.c_int64, C.c_int32]
        l.oracle_importance.restype = C.c_int
        l.oracle_importance.argtypes = [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_void_p]
        _lib = l
    return _lib


@dataclass
class Quant:
    res: OracleResult
    row_logp: np.ndarray
    keep_mask: np.ndarray

    def __getattr__(self, k):
        return getattr(self.res, k)


def quantify(row_ptr, col_idx, p, cut_off=1e-8, limit_order=20, n_threads=1, row_offset=0, want_rows=True):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float64)
    n_rows = len(row_ptr) - 1
    res = OracleResult()
    row_logp = np.empty(n_rows, dtype=np.float64) if want_rows else None
    mask = np.zeros((n_rows + 63) // 64, dtype=np.uint64) if want_rows else None
    rc = lib().oracle_quantify(n_rows, len(p), row_ptr.ctypes.data, col_idx.ctypes.data, row_offset, p.ctypes.data,
                               cut_off, limit_order, n_threads, C.byref(res),
                               row_logp.ctypes.data if want_rows else None, mask.ctypes.data if want_rows else None)
    if rc != 0:
        raise ValueError("oracle_quantify: invalid input")
    return Quant(res, row_logp, mask)


def importance(row_ptr, col_idx, p, cut_off=1e-8, limit_order=20):
    """contrib[j] = sum of P_i over kept cut sets containing event j (oracle_importance)."""
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.zeros(len(p), dtype=np.float64)
    rc = lib().oracle_importance(len(row_ptr) - 1, len(p), row_ptr.ctypes.data, col_idx.ctypes.data, p.ctypes.data,
                                 cut_off, limit_order, out.ctypes.data)
    if rc != 0:
        raise ValueError("oracle_importance: invalid input")
    return out


def product_domain(row_ptr, col_idx, p):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    col_idx = np.ascontiguousarray(col_idx, dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float64)
    a, b = C.c_double(), C.c_double()
    lib().oracle_quantify_product_domain(len(row_ptr) - 1, row_ptr.ctypes.data, col_idx.ctypes.data, p.ctypes.data,
                                         C.byref(a), C.byref(b))
    return a.value, b.value


def combine(parts):
    arr = (OracleResult * len(parts))(*parts)
    out = OracleResult()
    lib().oracle_combine(arr, len(parts), C.byref(out))
    return out


def synth_csr(seed, n_rows, n_events, avg_nnz, first_row=0, n_threads=1):
    nnz = lib().oracle_synth_nnz(seed, first_row, n_rows, avg_nnz)
    row_ptr = np.empty(n_rows + 1, dtype=np.int64)
    col_idx = np.empty(nnz, dtype=np.int32)
    rc = lib().oracle_synth_csr(seed, first_row, n_rows, n_events, avg_nnz, n_threads, row_ptr.ctypes.data,
                                col_idx.ctypes.data)
    if rc != 0:
        raise ValueError("oracle_synth_csr: invalid arguments")
    return row_ptr, col_idx


def synth_probabilities(seed, n_events):
    p = np.empty(n_events, dtype=np.float64)
    lib().oracle_synth_probabilities(seed, n_events, p.ctypes.data)
    return p


def shard_rows(row_ptr, n_shards):
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
    b = np.empty(n_shards + 1, dtype=np.int64)
    lib().oracle_shard_rows(row_ptr.ctypes.data, len(row_ptr) - 1, n_shards, b.ctypes.data)
    return b


def mc(dag, p, seed, num_trials, first_sample=0, n_threads=1):
    """dag: dict with n_events, gate_type, gate_min, arg_ptr, args, top (canopy_dag_view fields)."""
    gt = np.ascontiguousarray(dag["gate_type"], dtype=np.int32)
    gm = np.ascontiguousarray(dag["gate_min"], dtype=np.int32)
    ap = np.ascontiguousarray(dag["arg_ptr"], dtype=np.int32)
    ar = np.ascontiguousarray(dag["args"], dtype=np.int32)
    p = np.ascontiguousarray(p, dtype=np.float64)
    hits = lib().oracle_mc(dag["n_events"], len(gt), gt.ctypes.data, gm.ctypes.data, ap.ctypes.data, ar.ctypes.data,
                           dag["top"], p.ctypes.data, seed, first_sample, num_trials, n_threads)
    if hits < 0:
        raise ValueError("oracle_mc: invalid input")
    return int(hits)

"""Host-side mirror of the K0 entry points of include/canopy.h: Open-PSA MEF fault tree -> basic-event
probabilities, gate DAG (Monte-Carlo path) and minimal cut sets as CSR (quantification path).
Pure host logic inside libcanopy.so; usable without a GPU."""
import ctypes as C

import numpy as np

from . import _capi as capi


class FaultTree:
    def __init__(self, xml_text=None, path=None):
        self._h = C.c_void_p()
        err = C.create_string_buffer(512)
        if path is not None:
            st = capi.lib().canopy_ft_parse_file(str(path).encode(), C.byref(self._h), err, len(err))
        else:
            data = xml_text.encode() if isinstance(xml_text, str) else xml_text
            st = capi.lib().canopy_ft_parse(data, len(data), C.byref(self._h), err, len(err))
        if st != capi.OK:
            raise capi.CanopyError(st, "canopy_ft_parse", err.value.decode())
        self.info = capi.FtInfo()
        capi.check(capi.lib().canopy_ft_get_info(self._h, C.byref(self.info)), "canopy_ft_get_info")

    def close(self):
        if self._h:
            capi.lib().canopy_ft_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def basic_event_names(self):
        return [capi.lib().canopy_ft_basic_event_name(self._h, j).decode() for j in range(self.info.n_basic_events)]

    @property
    def top_gate_name(self):
        return capi.lib().canopy_ft_gate_name(self._h, self.info.top_gate).decode()

    def probabilities(self):
        p = np.empty(self.info.n_basic_events, dtype=np.float64)
        capi.check(capi.lib().canopy_ft_probabilities(self._h, p.ctypes.data), "canopy_ft_probabilities")
        return p

    def dag(self):
        """canopy_dag_view fields as a dict (for quantifier.Dag / the oracle)."""
        n = self.info.n_gates
        gt = np.empty(n, dtype=np.int32); gm = np.empty(n, dtype=np.int32)
        ap = np.empty(n + 1, dtype=np.int32); ar = np.empty(max(self.info.n_dag_args, 1), dtype=np.int32)
        top = C.c_int32()
        capi.check(capi.lib().canopy_ft_dag(self._h, gt.ctypes.data, gm.ctypes.data, ap.ctypes.data, ar.ctypes.data,
                                            C.byref(top)), "canopy_ft_dag")
        return dict(n_events=self.info.n_basic_events, gate_type=gt, gate_min=gm, arg_ptr=ap,
                    args=ar[:self.info.n_dag_args], top=top.value)

    def cut_sets(self, limit_order=0):
        """(row_ptr int64[n+1], col_idx int32[nnz]) -- the CSR canopy_model_upload takes."""
        n, nnz = C.c_int64(), C.c_int64()
        capi.check(capi.lib().canopy_ft_cut_sets(self._h, limit_order, C.byref(n), C.byref(nnz), None, None),
                   "canopy_ft_cut_sets")
        row_ptr = np.zeros(n.value + 1, dtype=np.int64)
        col = np.zeros(nnz.value, dtype=np.int32)
        capi.check(capi.lib().canopy_ft_cut_sets(self._h, limit_order, C.byref(n), C.byref(nnz), row_ptr.ctypes.data,
                                                 col.ctypes.data if nnz.value else None), "canopy_ft_cut_sets")
        return row_ptr, col

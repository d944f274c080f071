"""canopy_amd -- MI355X-native cut-set quantification behind the C ABI of include/canopy.h.

Only what the hot path needs lives here: csrc/ (hand-written HIP for gfx950 + the C ABI),
the ctypes binding (_capi), the host-side mirror of the ABI (quantifier) and the one-collective
multi-GPU combine (distributed). PyTorch is used for torch.distributed (RCCL) plumbing only.
"""
from . import _capi
from .quantifier import (Comm, Context, Model, QuantResult, combine_partials, default_params, importance_measures, shard_rows,
                         synthetic_probabilities, synthetic_shard_bounds)

__all__ = ["Comm", "synthetic_shard_bounds", "Context", "Model", "QuantResult", "combine_partials", "default_params", "importance_measures", "shard_rows",
           "synthetic_probabilities", "_capi"]

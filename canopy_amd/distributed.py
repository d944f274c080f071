"""K5 -- cross-GPU combine of the shard partials: ONE sum all-reduce (RCCL over xGMI).

Cut-set rows partition trivially across ranks (one process per GPU); the only exchange step of
the path is the combine of the per-shard partial records. Each rank's final-reduce kernel writes
its record into slot `rank` of a world_size x 8 buffer and zeroes the other slots
(canopy_ctx_set_gather_buffer), so a single `all_reduce(SUM)` is an all-gather of the records:
every rank then folds them in rank order with canopy_combine_partials -> bitwise identical results
on all ranks, independent of RCCL's reduction order (x + 0 is exact). The message is
world_size * 64 B: latency-bound, xGMI link bandwidth is irrelevant.

The same class runs on CPU tensors with the gloo backend (tests/test_distributed.py), where the
local record is put into the slot by the host instead of the kernel.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _capi as capi
from .quantifier import combine_partials


def library_comm(ctx, group=None):
    """Bootstrap libcanopy's own RCCL communicator (canopy_comm_init_rank) over an existing torch.distributed
    group of ANY backend (gloo is enough): rank 0 makes the 128-byte id, one broadcast hands it out. After this the
    data path needs no Python: canopy_quantify_async itself ends with the all-reduce. -> canopy_amd.Comm"""
    from .quantifier import Comm
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    box = [None]
    if rank == 0:
        try:
            box[0] = Comm.unique_id()
        except Exception as e:  # noqa: BLE001 -- every rank must learn of it: the others are waiting in the broadcast
            box[0] = ("error", repr(e))
    dist.broadcast_object_list(box, src=0, group=group)
    if isinstance(box[0], tuple):
        raise RuntimeError("canopy_comm_get_unique_id failed on rank 0: %s" % box[0][1])
    return Comm(ctx, box[0], world, rank)


class ShardCombiner:
    def __init__(self, ctx=None, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ctx = ctx
        if ctx is not None:
            device = torch.device("cuda", ctx.device) if device is None else device
        else:
            device = torch.device("cpu") if device is None else device
        self.buf = torch.zeros(self.world * capi.PARTIAL_DOUBLES, dtype=torch.float64, device=device)
        if ctx is not None:
            capi.check(capi.lib().canopy_ctx_set_gather_buffer(ctx.handle, self.buf.data_ptr(), self.world,
                                                               self.rank), "canopy_ctx_set_gather_buffer", ctx.handle)

    def detach(self):
        if self.ctx is not None:
            capi.lib().canopy_ctx_set_gather_buffer(self.ctx.handle, None, 0, 0)

    def put_local(self, partial):
        """CPU / test path: place this rank's record (capi.Partial) in its slot, zero the others."""
        rec = np.frombuffer(bytes(partial), dtype=np.float64).copy()
        self.buf.zero_()
        n = capi.PARTIAL_DOUBLES
        self.buf[self.rank * n:(self.rank + 1) * n] = torch.from_numpy(rec).to(self.buf.device)

    def all_reduce(self, force=False):
        """The single collective of the path. Enqueued on the current torch stream for RCCL: the
        caller makes the context's stream current (`with torch.cuda.stream(s)`), so the collective is
        ordered after the final-reduce kernel that fills this rank's slot."""
        if self.world > 1 or (force and dist.is_initialized()):
            dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=self.group)

    def partials(self):
        if self.ctx is not None:
            self.ctx.synchronize()
        host = self.buf.detach().to("cpu").numpy()
        n = capi.PARTIAL_DOUBLES
        return [capi.Partial(*host[k * n:(k + 1) * n].tolist()) for k in range(self.world)]

    def result(self, flags=0):
        return combine_partials(self.partials(), flags)


class BatchCombiner:
    """K5 for a batch (N4): every rank holds n_trials shard records; ONE all-gather of
    n_trials * 64 B per rank delivers all of them to all ranks, and each trial is folded over the
    ranks in rank order (canopy_combine_partials) -> identical results everywhere. On the GPU the
    trial kernels write straight into `local` (pass `device_out_ptr()` to quantify_batch_async)."""

    def __init__(self, n_trials, ctx=None, group=None, device=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ctx = ctx
        self.n_trials = n_trials
        if device is None:
            device = torch.device("cuda", ctx.device) if ctx is not None else torch.device("cpu")
        n = n_trials * capi.PARTIAL_DOUBLES
        self.local = torch.zeros(n, dtype=torch.float64, device=device)
        self.gathered = torch.zeros(self.world * n, dtype=torch.float64, device=device)

    def device_out_ptr(self):
        return self.local.data_ptr()

    def put_local(self, partials):
        """CPU / test path: this rank's n_trials records (sequence of capi.Partial)."""
        rec = np.concatenate([np.frombuffer(bytes(q), dtype=np.float64) for q in partials])
        self.local.copy_(torch.from_numpy(rec.copy()).to(self.local.device))

    def all_gather(self):
        """The single collective of the batch (same stream discipline as ShardCombiner.all_reduce)."""
        if self.world > 1:
            n = self.local.numel()
            dist.all_gather([self.gathered[r * n:(r + 1) * n] for r in range(self.world)], self.local, group=self.group)
        else:
            self.gathered.copy_(self.local)

    def results(self, flags=0):
        if self.ctx is not None:
            self.ctx.synchronize()
        host = self.gathered.detach().to("cpu").numpy().reshape(self.world, self.n_trials, capi.PARTIAL_DOUBLES)
        return [combine_partials([capi.Partial(*host[r, k].tolist()) for r in range(self.world)], flags)
                for k in range(self.n_trials)]

"""Device-memory leak check: contexts, models, every derived layout, communicators, DAGs and graphs created and freed in a
loop; the free device memory (hipMemGetInfo through torch) must come back to where it started."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import canopy_amd
from canopy_amd import _capi
SEED = 20250704

def cycle(k):
    ctx = canopy_amd.Context(device=0)
    m = ctx.synthetic(SEED + k, 1_500_000 + 1000 * k, 30_000, 6)
    m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, 30_000))
    for kern in (_capi.KERNEL_AUTO, _capi.KERNEL_ROWS, _capi.KERNEL_WAVE_STREAM, _capi.KERNEL_VALUE_SCATTER, _capi.KERNEL_BAND_STREAM):
        ctx.quantify(m, canopy_amd.default_params(kernel=kern, flags=_capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK))
    for kern in (_capi.KERNEL_FILTER_REFINE, _capi.KERNEL_FIXED_SUM):
        ctx.quantify(m, canopy_amd.default_params(kernel=kern))
    ctx.quantify(m, canopy_amd.default_params(flags=_capi.QUANT_IMPORTANCE))
    ts = [ctx.enqueue(m, canopy_amd.default_params()) for _ in range(8)]
    for t in ts:
        ctx.ticket_fetch(t)
    pt = np.clip(canopy_amd.synthetic_probabilities(SEED, 30_000)[None, :] * np.linspace(0.5, 1.5, 8)[:, None], 0, 1)
    ctx.quantify_batch(m, pt, canopy_amd.default_params())
    comm = canopy_amd.Comm(ctx, canopy_amd.Comm.unique_id(), 1, 0)
    ctx.quantify_async(m, canopy_amd.default_params()); comm.fetch()
    comm.close()
    g = ctx.graph_capture(m, canopy_amd.default_params())
    ctx.graph_launch(g); ctx.fetch_partial(); ctx.graph_free(g)
    m.close(); ctx.close()

free0 = None
for k in range(12):
    cycle(k)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    if k == 1:
        free0 = free  # (after the first cycles: the runtime's own pools are warm)
    print("cycle", k, "free MiB", free >> 20, flush=True)
leak = free0 - free
print("leak over 10 cycles: %.1f MiB" % (leak / 2**20))
sys.exit(1 if leak > 64 * 2**20 else 0)

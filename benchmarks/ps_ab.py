"""A/B of the part-sweep kernel's phases (CANOPY_ABLATIONS build, CANOPY_PS_DEBUG=<bits>): wrong results, timing only."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch, canopy_amd
from canopy_amd import _capi
ctx = canopy_amd.Context(device=0)
n, e = 100_000_000, 100_000
m = ctx.synthetic(20250704, n, e, 6); m.set_probabilities(canopy_amd.synthetic_probabilities(20250704, e))
m.prepare(_capi.KERNEL_PART_SWEEP); ctx.synchronize()
for flags, fl in ((0, "sum-only"), (_capi.QUANT_ROW_LOGP, "row_logp")):
    prm = canopy_amd.default_params(cut_off=1e-8, flags=flags, kernel=_capi.KERNEL_PART_SWEEP)
    for _ in range(3): ctx.quantify_async(m, prm)
    ctx.synchronize()
    ctx.timing_begin(10)
    for _ in range(10): ctx.quantify_async(m, prm)
    ks, ts, nn = ctx.timing_end()
    print(fl, "kernel ms %.4f" % (ks / nn), flush=True)

import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, canopy_amd
from canopy_amd import _capi
SEED=20250704
ctx = canopy_amd.Context(device=0)
n, e = 100_000_000, 100_000
m = ctx.synthetic(SEED, n, e, 6); m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
for kern in ((_capi.KERNEL_BAND_STREAM,) if os.environ.get("BS_ONLY") else (_capi.KERNEL_BAND_STREAM, _capi.KERNEL_WAVE_STREAM)):
    for name, flags, co in (("rows+mask", _capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK, 1e-8), ("rows", _capi.QUANT_ROW_LOGP, 1e-8), ("sum-only", 0, 1e-8), ("rows cut_off 0", _capi.QUANT_ROW_LOGP, 0.0), ("sum-only cut_off 1 (nothing kept)", 0, 1.0)):
        prm = canopy_amd.default_params(cut_off=co, flags=flags, kernel=kern)
        for _ in range(3): ctx.quantify_async(m, prm)
        ctx.synchronize()
        ctx.timing_begin(10)
        for _ in range(10): ctx.quantify_async(m, prm)
        ctx.synchronize()
        _,_,k = ctx.timing_end(); l = ctx.timing_launches(k)
        ks = sorted(a for a,_ in l)
        print("kernel", kern, "%-36s median ms %.4f min %.4f" % (name, ks[len(ks)//2], ks[0]), flush=True)

import sys, os
sys.path.insert(0, os.getcwd())
import canopy_amd
from canopy_amd import _capi
SEED=20250704
ctx = canopy_amd.Context(device=0)
for e, avg in ((1000, 4), (1000, 6), (2048, 6), (8000, 6), (12288, 6)):
    n = 100_000_000
    m = ctx.synthetic(SEED, n, e, avg); m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
    out = []
    for kern in (_capi.KERNEL_ROWS_LDS, _capi.KERNEL_BAND_STREAM):
        for flags, name in ((_capi.QUANT_ROW_LOGP, "rows"), (0, "sum")):
            prm = canopy_amd.default_params(cut_off=1e-8, flags=flags, kernel=kern)
            for _ in range(3): ctx.quantify_async(m, prm)
            ctx.synchronize()
            ctx.timing_begin(10)
            for _ in range(10): ctx.quantify_async(m, prm)
            ctx.synchronize()
            _,_,k = ctx.timing_end(); l = ctx.timing_launches(k)
            ks = sorted(a for a,_ in l)
            out.append("%d/%s: %.4f" % (kern, name, ks[len(ks)//2]))
    print("E", e, "avg", avg, "  ".join(out), flush=True)
    m.close()

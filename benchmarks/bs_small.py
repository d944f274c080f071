import sys, os
sys.path.insert(0, os.getcwd())
import canopy_amd
from canopy_amd import _capi
SEED=20250704
ctx = canopy_amd.Context(device=0)
for e in (100_000, 10_000):
    for n in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000):
        m = ctx.synthetic(SEED, n, e, 6); m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
        out = []
        for kern in (_capi.KERNEL_ROWS, _capi.KERNEL_WAVE_STREAM, _capi.KERNEL_BAND_STREAM):
            prm = canopy_amd.default_params(cut_off=1e-8, flags=_capi.QUANT_ROW_LOGP, kernel=kern)
            for _ in range(3): ctx.quantify_async(m, prm)
            ctx.synchronize()
            ctx.timing_begin(20)
            for _ in range(20): ctx.quantify_async(m, prm)
            ctx.synchronize()
            _,_,k = ctx.timing_end(); l = ctx.timing_launches(k)
            ks = sorted(a for a,_ in l)
            out.append("%d: %.4f" % (kern, ks[len(ks)//2]))
        print("E", e, "rows", n, "  ".join(out), flush=True)
        m.close()

import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np, canopy_amd, oracle_py
from canopy_amd import _capi
SEED=20250704
ctx = canopy_amd.Context(device=0)
for n, e in ((3_000_000, 100_000), (100_000_000, 100_000)):
    m = ctx.synthetic(SEED, n, e, 6); m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
    res = {}
    for kern in (_capi.KERNEL_WAVE_STREAM, _capi.KERNEL_BAND_STREAM):
        prm = canopy_amd.default_params(cut_off=1e-8, flags=_capi.QUANT_ROW_LOGP | _capi.QUANT_KEEP_MASK, kernel=kern)
        t0=time.time(); g = ctx.quantify(m, prm); tb=time.time()-t0
        rows = ctx.fetch_row_logp(m); mask = ctx.fetch_keep_mask(m)
        for _ in range(3): ctx.quantify_async(m, prm)
        ctx.synchronize()
        ctx.timing_begin(10)
        for _ in range(10): ctx.quantify_async(m, prm)
        ctx.synchronize()
        _,_,k = ctx.timing_end(); l = ctx.timing_launches(k)
        ks = sorted(a for a,_ in l)
        res[kern] = (g, rows, mask)
        print(n, "kernel", kern, "ran", ctx.last_kernel_variant(), "median ms %.4f min %.4f first-call s %.2f" % (ks[len(ks)//2], ks[0], tb), g.n_kept, g.argmax_row, g.rare_event, flush=True)
    a, b = res[_capi.KERNEL_WAVE_STREAM], res[_capi.KERNEL_BAND_STREAM]
    print("rows identical:", np.array_equal(a[1], b[1]), "mask identical:", np.array_equal(a[2], b[2]),
          "records identical:", (a[0].n_kept, a[0].argmax_row, a[0].rare_event, a[0].mcub, a[0].max_logp) == (b[0].n_kept, b[0].argmax_row, b[0].rare_event, b[0].mcub, b[0].max_logp))
    info = m.info(); print("stream_bytes_exact", info.stream_bytes_exact, "per row %.2f" % (info.stream_bytes_exact / n))
    m.close()

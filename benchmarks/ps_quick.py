import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import torch, canopy_amd, oracle_py
from canopy_amd import _capi
SEED=20250704
ctx = canopy_amd.Context(device=0)
def check(n, e, avg, cut):
    rp, col = oracle_py.synth_csr(SEED, n, e, avg, n_threads=16)
    p = oracle_py.synth_probabilities(SEED, e)
    want = oracle_py.quantify(rp, col, p, cut_off=cut, n_threads=16)
    m = ctx.upload(rp, col, e); m.set_probabilities(p)
    prm = canopy_amd.default_params(cut_off=cut, flags=_capi.QUANT_ROW_LOGP|_capi.QUANT_KEEP_MASK, kernel=_capi.KERNEL_PART_SWEEP)
    got = ctx.quantify(m, prm)
    rows = ctx.fetch_row_logp(m); mask = ctx.fetch_keep_mask(m)
    ok = np.array_equal(rows, want.row_logp) and np.array_equal(mask, want.keep_mask) and got.n_kept==int(want.n_kept) and got.argmax_row==int(want.argmax_row)
    print("check", n, e, avg, cut, "OK" if ok else "MISMATCH", got.n_kept, int(want.n_kept), "bad rows", int((rows!=want.row_logp).sum()), flush=True)
    m.close()
    return ok
allok = True
for (n,e,avg,cut) in [(1000,500,4,1e-8),(5000,9000,6,0.0),(100000,100000,6,1e-8),(300001,20000,6,1e-9),(70000,262144,6,0.0),(8193,8192,3,1e-8)]:
    allok &= check(n,e,avg,cut)
print("ALL OK" if allok else "FAILED", flush=True)
if allok and len(sys.argv) > 1:
    n, e = 100_000_000, 100_000
    m = ctx.synthetic(SEED, n, e, 6); m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, e))
    for kern, name in ((_capi.KERNEL_PART_SWEEP,"PS"),(_capi.KERNEL_WAVE_STREAM,"WS")):
        t0=time.time(); m.prepare(kern); ctx.synchronize(); print(name, "prepare s", time.time()-t0, "stream_bytes", m.info().stream_bytes, flush=True)
        for flags, fl in ((0,"sum-only"),(_capi.QUANT_ROW_LOGP,"row_logp")):
            for cut in (1e-8, 0.0):
                prm = canopy_amd.default_params(cut_off=cut, flags=flags, kernel=kern)
                for _ in range(3): ctx.quantify_async(m, prm)
                ctx.synchronize()
                ctx.timing_begin(10)
                for _ in range(10): ctx.quantify_async(m, prm)
                ks, ts, nn = ctx.timing_end()
                print(name, fl, "cut", cut, "kernel ms %.4f pass ms %.4f" % (ks/nn, ts/nn), flush=True)

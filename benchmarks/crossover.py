"""Where do the two-step passes overtake the row kernels? Explicit variants on small models: serial passes (each fetched before the
next) and pipelined sequences (tickets, four lanes), cut_off 1e-8, sum-only. usage (via gpurun): python benchmarks/crossover.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import canopy_amd
from canopy_amd import _capi
SEED = 20250704
ctx = canopy_amd.Context(device=0)
def run(m, kern, pipelined, n=300):
    prm = canopy_amd.default_params(kernel=kern)
    ctx.set_pipeline(4 if pipelined else 0)
    def loop(k):
        infl = []
        for _ in range(k):
            infl.append(ctx.enqueue(m, prm))
            if len(infl) == (4 if pipelined else 1): ctx.ticket_fetch(infl.pop(0))
        while infl: ctx.ticket_fetch(infl.pop(0))
    loop(40); ctx.synchronize()
    t0 = time.perf_counter(); loop(n); ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e6
for e in (1000, 100_000):
    p = canopy_amd.synthetic_probabilities(SEED, e)
    for rows in (50_000, 100_000, 200_000, 400_000, 800_000, 1_600_000):
        m = ctx.synthetic(SEED, rows, e, 6); m.set_probabilities(p)
        base = _capi.KERNEL_ROWS_LDS if e <= 2048 else _capi.KERNEL_ROWS
        out = []
        for kern, name in ((base, "rows"), (_capi.KERNEL_FILTER_REFINE, "filter/refine")):
            m.prepare(kern)
            out.append("%s serial %.1f us pipelined %.1f us" % (name, run(m, kern, False), run(m, kern, True)))
        print("E %6d rows %8d: %s" % (e, rows, "; ".join(out)), flush=True)
        m.close()

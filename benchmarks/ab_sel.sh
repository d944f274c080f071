#!/bin/bash
# A/B of two library builds (benchmarks/bin/libcanopy_A.so / _B.so) on the legs the exact pass matters for: the serial pass, the
# 8-way shard step and the pass at cut-offs more rows survive. usage (via gpurun): bash benchmarks/ab_sel.sh
set -u
for rep in 1 2; do for v in A B; do
CANOPY_AMD_LIB=$PWD/benchmarks/bin/libcanopy_$v.so python3 - <<PY
import sys, os, time
sys.path.insert(0, os.getcwd())
import canopy_amd
SEED=20250704
ctx = canopy_amd.Context(device=0)
m = ctx.synthetic(SEED, 100_000_000, 100_000, 6); m.set_probabilities(canopy_amd.synthetic_probabilities(SEED, 100_000))
out=[]
for co in (1e-8, 1e-10, 1e-12):
    prm = canopy_amd.default_params(cut_off=co)
    ctx.set_pipeline(0)
    for _ in range(5): ctx.quantify_async(m, prm)
    ctx.fetch_partial()
    ctx.timing_begin(20)
    for _ in range(20): ctx.quantify_async(m, prm)
    ctx.synchronize()
    _,_,k = ctx.timing_end(); l = ctx.timing_launches(k); ks = sorted(a for a,_ in l)
    ctx.set_pipeline(4)
    for _ in range(8): ctx.quantify_async(m, prm)
    ctx.synchronize()
    t0=time.time(); ts=[]
    for i in range(100):
        ts.append(ctx.enqueue(m, prm))
        if len(ts)==4: ctx.ticket_fetch(ts.pop(0))
    for t in ts: ctx.ticket_fetch(t)
    ctx.synchronize(); dt=(time.time()-t0)/100
    out.append("co %g: kernels %.4f step %.4f" % (co, ks[len(ks)//2], dt*1e3))
print("$v", " | ".join(out), flush=True)
PY
done; done

"""Eager launches vs a captured pass (canopy_graph_capture / canopy_graph_launch) on the launch-bound config-2 model and on
config 3. Measured on MI355X: 22.99 vs 22.75 us per step (1e5 cut sets), 472.5 vs 470.3 us (1e8): the small model's floor is
device-side dependency latency between its two kernels, not host launch cost."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import canopy_amd
from canopy_amd import _capi
SEED=20250704
ctx = canopy_amd.Context(device=0)
for (n,e,avg) in ((100_000,1000,4),(100_000_000,100_000,6)):
    p = canopy_amd.synthetic_probabilities(SEED, e)
    m = ctx.synthetic(SEED, n, e, avg); m.set_probabilities(p)
    prm = canopy_amd.default_params()
    want = ctx.quantify(m, prm)
    g = ctx.graph_capture(m, prm)
    for mode in ("eager","graph"):
        for _ in range(20):
            ctx.quantify_async(m, prm) if mode=="eager" else ctx.graph_launch(g)
        ctx.synchronize()
        t0=time.perf_counter(); K=200 if n<1e6 else 40
        for _ in range(K):
            ctx.quantify_async(m, prm) if mode=="eager" else ctx.graph_launch(g)
        ctx.synchronize()
        dt=(time.perf_counter()-t0)/K
        got = canopy_amd.combine_partials([ctx.fetch_partial()])
        print(n, mode, "us/step %.2f" % (dt*1e6), got.n_kept==want.n_kept, got.rare_event==want.rare_event)
    # p changes between replays
    p2 = p*0.5; m.set_probabilities(p2); ctx.graph_launch(g); a = canopy_amd.combine_partials([ctx.fetch_partial()]); b = ctx.quantify(m, prm)
    print("after p change:", a.n_kept==b.n_kept, a.rare_event==b.rare_event)
    ctx.graph_free(g); m.close()

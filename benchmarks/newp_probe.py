"""Where does a new probability vector per step cost time? (config 3, one GPU)"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, canopy_amd
SEED = 20250704
ctx = canopy_amd.Context(device=0)
n, e = 100_000_000, 100_000
m = ctx.synthetic(SEED, n, e, 6)
p = canopy_amd.synthetic_probabilities(SEED, e)
m.set_probabilities(p); m.prepare(0); ctx.synchronize()
prm = canopy_amd.default_params()
lp = np.log(p)
def loop(n_steps, pre, lanes=4):
    ctx.set_pipeline(lanes)
    infl = []
    t_host = 0.0
    for i in range(n_steps):
        t0 = time.perf_counter(); pre(i); t_host += time.perf_counter() - t0
        infl.append(ctx.enqueue(m, prm))
        if len(infl) == lanes: ctx.ticket_fetch(infl.pop(0))
    while infl: ctx.ticket_fetch(infl.pop(0))
    return t_host
for lanes in (4, 3):
    for name, pre in (("same p, no set", lambda i: None), ("set_log (same values)", lambda i: m.set_log_probabilities(lp)),
                      ("set_p (same values)", lambda i: m.set_probabilities(p))):
        loop(40, pre, lanes); ctx.synchronize()
        t0 = time.perf_counter(); th = loop(200, pre, lanes); ctx.synchronize(); el = time.perf_counter() - t0
        print("lanes %d  %-24s %.4f ms/step   host in set: %.4f ms/step" % (lanes, name, el * 5, th * 5), flush=True)

"""Soak of the row stream on mid-size synthetic models: every bit class (12 .. 18), both tables, one / two / four slabs per step,
pipelined sequences and passes alone, cut-offs from rare to generous -- every record against the oracle (integers exact, sums 1e-12).
usage (via gpurun): python benchmarks/soak_ls.py [n_models]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import canopy_amd, oracle_py
from canopy_amd import _capi
n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(20251005)
ctx = canopy_amd.Context(device=0)
cores = len(os.sched_getaffinity(0))
bad = 0
for k in range(n_models):
    e = int(rng.choice([900, 4000, 12000, 40000, 66000, 100000, 125000, 180000, 255000]))
    rows = int(rng.choice([70_000, 250_000, 600_000, 1_300_000, 2_700_000, 5_000_000]))
    avg = int(rng.choice([3, 4, 6, 9]))
    seed = int(rng.integers(1, 1 << 30))
    rp, col = oracle_py.synth_csr(seed, rows, e, avg, n_threads=cores)
    p = oracle_py.synth_probabilities(seed, e)
    if rng.random() < 0.5:
        p = np.minimum(1.0, p * 10.0 ** rng.uniform(0, 2.5))
    co = float(rng.choice([1e-10, 1e-8, 1e-6, 1e-4]))
    lo = int(rng.choice([0, 20, 20, 5]))
    w = oracle_py.quantify(rp, col, p, cut_off=co, limit_order=lo, n_threads=cores, want_rows=False)
    m = ctx.upload(rp, col, e); m.set_probabilities(p)
    prm = canopy_amd.default_params(cut_off=co, limit_order=lo, kernel=_capi.KERNEL_FILTER_REFINE)
    def ok(g):
        return (g.n_kept == int(w.n_kept) and g.argmax_row == int(w.argmax_row) and (not int(w.n_kept) or g.max_logp == w.max_logp)
                and abs(g.rare_event - w.rare_event) <= 1e-12 * abs(w.rare_event) and abs(g.mcub - w.mcub) <= 1e-12 * abs(w.mcub))
    res = []
    ctx.set_pipeline(0)
    res.append(ok(ctx.quantify(m, prm)))                      # a pass alone
    ctx.set_pipeline(4)
    tk = [ctx.enqueue(m, prm) for _ in range(6)]               # a pipelined sequence
    res += [ok(canopy_amd.combine_partials([ctx.ticket_fetch(t)])) for t in tk]
    prm_m = canopy_amd.default_params(cut_off=co, limit_order=lo, kernel=_capi.KERNEL_FILTER_REFINE, flags=_capi.QUANT_KEEP_MASK)
    g = ctx.quantify(m, prm_m)
    wm = oracle_py.quantify(rp, col, p, cut_off=co, limit_order=lo, n_threads=cores, want_rows=True).keep_mask
    res.append(ok(g) and np.array_equal(ctx.fetch_keep_mask(m), wm))
    info = m.info()
    print("model %2d: E %6d rows %8d avg %d cut_off %.0e limit %2d  rows stream %s  kept %9d  %s" % (
        k, e, rows, avg, co, lo, "yes" if info.stream_bytes_rows else "NO ", int(w.n_kept), "ok" if all(res) else "MISMATCH %s" % res), flush=True)
    bad += not all(res)
    m.close()
print("soak: %d models, %d mismatches" % (n_models, bad))
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""A model family the synthetic generator does not cover: HOT basic events (a few events in most cut sets, as in real fault
trees), uploaded from the host. Checks AUTO at cut_off 1e-8 and 0 against the oracle and times the passes.
usage: python benchmarks/bench_skewed.py [rows] [events] [hot_events] [hot_share]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch  # noqa: F401
import canopy_amd
import oracle_py

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
hot = int(sys.argv[3]) if len(sys.argv) > 3 else 50
share = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
rng = np.random.default_rng(7)
L = 6
# every row: L distinct columns; each slot is a hot event with probability `share` (duplicates re-drawn from the cold range)
cols = rng.integers(hot, E, size=(rows, L), dtype=np.int64)
is_hot = rng.random((rows, L)) < share
cols[is_hot] = rng.integers(0, hot, size=int(is_hot.sum()))
cols.sort(axis=1)
dup = np.zeros((rows, L), dtype=bool)
dup[:, 1:] = cols[:, 1:] == cols[:, :-1]
keep = ~dup
lens = keep.sum(axis=1)
row_ptr = np.zeros(rows + 1, dtype=np.int64); row_ptr[1:] = np.cumsum(lens)
col = cols[keep].astype(np.int32)
p = 10.0 ** rng.uniform(-4, -1, size=E)
p[:hot] = 10.0 ** rng.uniform(-2, -0.5, size=hot)  # hot events are also the likely ones
cores = len(os.sched_getaffinity(0))
ctx = canopy_amd.Context(device=0)
t0 = time.time(); m = ctx.upload(row_ptr, col, E); m.set_probabilities(p); m.prepare(0); ctx.synchronize(); t_up = time.time() - t0
out = {"rows": rows, "n_events": E, "hot_events": hot, "hot_share": share, "nnz": int(row_ptr[-1]), "upload_prepare_s": t_up}
for cut in (1e-8, 0.0):
    prm = canopy_amd.default_params(cut_off=cut)
    got = ctx.quantify(m, prm)
    for _ in range(10):
        ctx.quantify_async(m, prm)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        ctx.quantify_async(m, prm)
    ctx.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    w = oracle_py.quantify(row_ptr, col, p, cut_off=cut, n_threads=cores, want_rows=False)
    ok = got.n_kept == int(w.n_kept) and got.argmax_row == int(w.argmax_row) and got.max_logp == w.max_logp and \
        abs(got.rare_event - w.rare_event) <= 1e-12 * w.rare_event and abs(got.mcub - w.mcub) <= 1e-12 * abs(w.mcub)
    out["cut_off_%g" % cut] = {"ms_per_step": ms, "variant": ctx.last_kernel_variant(), "n_kept": got.n_kept,
                              "kept_fraction": got.n_kept / rows, "rare_event": got.rare_event, "matches_oracle": bool(ok),
                              "ns_per_row": ms * 1e6 / rows}
print(json.dumps(out))

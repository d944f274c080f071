#!/usr/bin/env python3
"""bench_batch.py -- N4: K probability vectors (uncertainty trials) against the config-3 model on one MI355X.

Compares canopy_quantify_batch (one envelope pass + K exact passes over the compacted candidates) with K
separate canopy_quantify passes (each a full stream of the model). Trials: log-normal multipliers
exp(sigma * z) around the synthetic point probabilities, z from numpy's seeded generator.
Prints one JSON line per (K, sigma).  usage: python benchmarks/bench_batch.py [--rows N] [--trials 64,256]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 20250704


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--events", type=int, default=100_000)
    ap.add_argument("--avg-nnz", type=int, default=6)
    ap.add_argument("--trials", default="64,256")
    ap.add_argument("--sigmas", default="0.1,0.5")
    ap.add_argument("--check", type=int, default=4, help="trials re-run as single passes and compared")
    args = ap.parse_args()
    import numpy as np
    import torch  # noqa: F401  (first: one HIP runtime per process)
    import canopy_amd

    ctx = canopy_amd.Context(device=0)
    p = canopy_amd.synthetic_probabilities(SEED, args.events)
    model = ctx.synthetic(SEED, args.rows, args.events, args.avg_nnz)
    model.set_probabilities(p)
    prm = canopy_amd.default_params()
    model.prepare(0)
    for _ in range(60):  # spin-up: the first ~20 launches after the idle model build run through a clock transient
        ctx.quantify_async(model, prm)
    ctx.synchronize()
    single_ms = float(np.median([ctx.quantify(model, prm).elapsed_kernel_ms for _ in range(9)]))
    for sigma in [float(x) for x in args.sigmas.split(",")]:
        for k in [int(x) for x in args.trials.split(",")]:
            rng = np.random.default_rng(1234)
            pt = np.minimum(1.0, p[None, :] * np.exp(sigma * rng.standard_normal((k, args.events))))
            logs = np.log(pt)
            # warm-up with the SAME trials: the compact-CSR buffers grow (a synchronous hipFree + hipMalloc between the batch's
            # events) the first time a batch lists more candidates than any before it -- round 2 warmed up with 8 trials and its
            # first 64-trial batch at sigma 0.5 (2.0e6 candidates after 1.1e6) carried that growth: candidate_ms 1.55 instead of 0.6
            ctx.quantify_batch(model, logs, prm, log_input=True)
            t0 = time.perf_counter()
            res, info = ctx.quantify_batch(model, logs, prm, log_input=True)
            wall_log_input = time.perf_counter() - t0
            t0 = time.perf_counter()
            res2, info2 = ctx.quantify_batch(model, pt, prm)
            wall_p_input = time.perf_counter() - t0
            ok = True
            for j in range(min(args.check, k)):
                model.set_probabilities(pt[j])
                s = ctx.quantify(model, prm)
                ok = ok and (s.n_kept, s.argmax_row) == (res[j].n_kept, res[j].argmax_row) \
                    and abs(s.rare_event - res[j].rare_event) <= 1e-12 * s.rare_event \
                    and (res2[j].n_kept, res2[j].rare_event) == (res[j].n_kept, res[j].rare_event)
            model.set_probabilities(p)
            dev_ms = info.candidate_ms + info.trials_ms
            print(json.dumps({
                "bench": "batched quantification", "rows": args.rows, "n_events": args.events, "n_trials": k, "sigma": sigma,
                "n_candidates": info.n_candidates, "candidate_fraction": info.n_candidates / args.rows,
                "candidate_ms": info.candidate_ms, "trials_ms": info.trials_ms, "device_ms": dev_ms,
                "device_ms_per_trial": dev_ms / k, "single_pass_ms": single_ms,
                "speedup_vs_k_single_passes_device": single_ms * k / dev_ms,
                "cut_set_quantifications_per_s_device": args.rows * k / (dev_ms * 1e-3),
                "wall_s_log_input": wall_log_input, "wall_s_p_input": wall_p_input, "host_log_ms_p_input": info2.host_log_ms,
                "matches_single_passes": bool(ok),
            }), flush=True)
    model.close()
    ctx.close()


if __name__ == "__main__":
    main()

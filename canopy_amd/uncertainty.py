"""Uncertainty analysis on top of canopy_quantify_batch (host-side convenience; N4).

The reference declares the deviates that drive it -- uniform, normal, log-normal (mean + error factor at a
confidence level, or mu/sigma), gamma, beta, histogram (src/mef/openpsa/expr/random_deviate.h:46-256; bodies
absent) -- on ONE shared std::mt19937 ("not suitable for parallelized simulations", :20-24) and the switches
Settings::uncertainty_analysis / num_trials / seed (settings.h:118-130,281-295,324-325). Here the trial matrix is
drawn on the host with numpy's seeded generator (same parametrisations, restated from the header's documentation),
clipped to [0, 1], and the device quantifies all trials in one batch; the statistics are plain numpy.
"""
import math

import numpy as np

from .quantifier import default_params

_Z = {0.90: 1.2815515655446004, 0.95: 1.6448536269514722, 0.99: 2.3263478740408408}


def _z(level):
    """z_alpha of the header's EF = exp(z_alpha * sigma) (random_deviate.h:109-111): the ONE-sided upper normal
    quantile, P(X < z) = level, as in the Open-PSA / SCRAM lineage -- the error factor is the `level` quantile of the
    deviate over its median (0.95 -> 1.645). The reference holds no body for this deviate, so the reading is
    unpinned by the reference."""
    if level in _Z:
        return _Z[level]
    if not 0.5 < level < 1.0:
        raise ValueError("confidence level must lie in (0.5, 1)")
    from statistics import NormalDist
    return NormalDist().inv_cdf(level)


def sample_trials(distributions, n_trials, seed=0):
    """distributions: one spec per basic event --
         float | ("constant", p) | ("uniform", lo, hi) | ("normal", mean, sigma) |
         ("lognormal", mean, error_factor[, level=0.95]) | ("lognormal-mu-sigma", mu, sigma) |
         ("gamma", k, theta) | ("beta", alpha, beta) | ("histogram", boundaries, weights)
       -> float64 [n_trials, n_events], clipped to [0, 1] (a probability)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_trials, len(distributions)), dtype=np.float64)
    for j, d in enumerate(distributions):
        if isinstance(d, (int, float)):
            d = ("constant", float(d))
        kind = d[0]
        if kind == "constant":
            col = np.full(n_trials, d[1])
        elif kind == "uniform":
            if not d[1] < d[2]:
                raise ValueError("uniform: min must be below max")
            col = rng.uniform(d[1], d[2], n_trials)
        elif kind == "normal":
            if not d[2] > 0:
                raise ValueError("normal: sigma must be positive")
            col = rng.normal(d[1], d[2], n_trials)
        elif kind == "lognormal":  # mean of the log-normal itself + error factor at a confidence level
            mean, ef = d[1], d[2]
            level = d[3] if len(d) > 3 else 0.95
            if not (mean > 0 and ef > 1):
                raise ValueError("lognormal: need mean > 0 and error factor > 1")
            sigma = math.log(ef) / _z(level)
            mu = math.log(mean) - 0.5 * sigma * sigma  # E(x) = exp(mu + sigma^2 / 2)
            col = rng.lognormal(mu, sigma, n_trials)
        elif kind == "lognormal-mu-sigma":
            if not d[2] > 0:
                raise ValueError("lognormal: sigma must be positive")
            col = rng.lognormal(d[1], d[2], n_trials)
        elif kind == "gamma":
            if not (d[1] > 0 and d[2] > 0):
                raise ValueError("gamma: k and theta must be positive")
            col = rng.gamma(d[1], d[2], n_trials)
        elif kind == "beta":
            if not (d[1] > 0 and d[2] > 0):
                raise ValueError("beta: alpha and beta must be positive")
            col = rng.beta(d[1], d[2], n_trials)
        elif kind == "histogram":
            b, w = np.asarray(d[1], dtype=np.float64), np.asarray(d[2], dtype=np.float64)
            if len(b) != len(w) + 1 or np.any(np.diff(b) <= 0) or np.any(w < 0) or w.sum() <= 0:
                raise ValueError("histogram: boundaries strictly increasing, one non-negative weight per interval")
            k = rng.choice(len(w), size=n_trials, p=w / w.sum())
            col = rng.uniform(b[k], b[k + 1])
        else:
            raise ValueError("unknown distribution %r" % (kind,))
        out[:, j] = col
    return np.clip(out, 0.0, 1.0)


def summarize(values, quantiles=(0.05, 0.5, 0.95)):
    v = np.asarray(values, dtype=np.float64)
    return {"mean": float(v.mean()), "std": float(v.std(ddof=1)) if v.size > 1 else 0.0,
            "quantiles": {q: float(np.quantile(v, q)) for q in quantiles}, "n_trials": int(v.size)}


def uncertainty_analysis(ctx, model, distributions, n_trials=1000, seed=0, params=None):
    """-> dict: per-trial results, statistics of the rare-event and MCUB top-event probability, batch info.
    n_trials / seed defaults are the reference's (settings.h:324-325)."""
    p_trials = sample_trials(distributions, n_trials, seed)
    results, info = ctx.quantify_batch(model, p_trials, params if params is not None else default_params())
    return {"results": results, "p_trials": p_trials,
            "rare_event": summarize([r.rare_event for r in results]), "mcub": summarize([r.mcub for r in results]),
            "n_candidates": info.n_candidates, "device_ms": info.candidate_ms + info.trials_ms}

"""Host-side uncertainty layer (canopy_amd/uncertainty.py): the deviate parametrisations against their closed-form
moments (CPU), and the end-to-end analysis on the device against per-trial oracle runs (-m gpu)."""
import math

import numpy as np
import pytest

SEED = 20250704


def test_deviate_parametrisations_have_the_documented_moments(canopy_lib):
    from canopy_amd.uncertainty import sample_trials, summarize
    n = 400_000
    t = sample_trials([0.25, ("uniform", 0.1, 0.3), ("normal", 0.5, 0.05), ("lognormal", 1e-3, 3.0), ("lognormal", 1e-2, 10.0, 0.90),
                       ("lognormal-mu-sigma", math.log(1e-3), 0.5), ("gamma", 2.0, 0.01), ("beta", 2.0, 8.0),
                       ("histogram", [0.0, 0.1, 0.3], [1.0, 3.0])], n, seed=5)
    assert t.shape == (n, 9) and t.min() >= 0.0 and t.max() <= 1.0
    m = t.mean(axis=0)
    assert np.all(t[:, 0] == 0.25)
    assert abs(m[1] - 0.2) < 1e-3 and abs(m[2] - 0.5) < 1e-3
    assert abs(m[3] - 1e-3) < 2e-5  # the mean parameter is the mean of the log-normal itself
    # error factor = quantile(level) / median: z_alpha is the one-sided upper quantile (random_deviate.h:109-111)
    med, hi = np.quantile(t[:, 3], [0.5, 0.95])
    assert abs(hi / med - 3.0) < 0.05
    med, hi = np.quantile(t[:, 4], [0.5, 0.90])
    assert abs(hi / med - 10.0) < 0.3
    assert abs(np.median(t[:, 5]) - 1e-3) < 2e-5
    assert abs(m[6] - 0.02) < 2e-4 and abs(m[7] - 0.2) < 1e-3
    assert abs(m[8] - (0.25 * 0.05 + 0.75 * 0.2)) < 1e-3
    s = summarize(t[:, 1])
    assert s["n_trials"] == n and s["quantiles"][0.05] < s["quantiles"][0.5] < s["quantiles"][0.95]
    with pytest.raises(ValueError):
        sample_trials([("lognormal", 1e-3, 0.5)], 10)
    with pytest.raises(ValueError):
        sample_trials([("histogram", [0.0, 0.1], [1.0, 2.0])], 10)


@pytest.mark.gpu
def test_uncertainty_analysis_end_to_end(gpu_ctx, oracle):
    import canopy_amd
    from canopy_amd.uncertainty import uncertainty_analysis
    n_rows, n_events, n_trials = 50_000, 400, 40
    row_ptr, col = oracle.synth_csr(SEED, n_rows, n_events, 4, n_threads=2)
    p = canopy_amd.synthetic_probabilities(SEED, n_events)
    dists = [("lognormal", float(pj), 3.0) if j % 3 else float(pj) for j, pj in enumerate(p)]
    model = gpu_ctx.upload(row_ptr, col, n_events)
    out = uncertainty_analysis(gpu_ctx, model, dists, n_trials=n_trials, seed=9)
    rare = []
    for k in range(n_trials):
        want = oracle.quantify(row_ptr, col, out["p_trials"][k], want_rows=False)
        got = out["results"][k]
        assert got.n_kept == int(want.n_kept) and got.argmax_row == int(want.argmax_row)
        assert abs(got.rare_event - want.rare_event) <= 1e-12 * want.rare_event
        rare.append(want.rare_event)
    assert abs(out["rare_event"]["mean"] - float(np.mean(rare))) <= 1e-12 * np.mean(rare)
    assert out["rare_event"]["quantiles"][0.05] <= out["rare_event"]["quantiles"][0.95] and out["mcub"]["std"] > 0
    model.close()

"""Importance contributions (N4, second half): contrib[j] = sum of P_i over kept cut sets containing event j.
CPU: the oracle against closed forms. GPU (-m gpu): the device pass against the oracle (<= 1e-12 relative per
event; the zero pattern exact) for every kernel variant AUTO can pick. Parity is unpinned by the reference
(only the `importance_analysis` flag exists: src/mef/openpsa/settings.h:262-279)."""
import numpy as np
import pytest

SEED = 20250704
TOL = 1e-12


def csr(rows):
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum([len(r) for r in rows])
    return row_ptr, np.array([c for r in rows for c in r], dtype=np.int32)


def test_oracle_importance_closed_forms(oracle, canopy_lib):
    from canopy_amd import importance_measures
    # 2-of-3 voting tree {ab, ac, bc} plus a single-event cut set {d}
    row_ptr, col = csr([[0, 1], [0, 2], [1, 2], [3]])
    p = np.array([0.1, 0.2, 0.3, 0.05])
    c = oracle.importance(row_ptr, col, p, cut_off=0.0)
    want = np.array([0.1 * 0.2 + 0.1 * 0.3, 0.1 * 0.2 + 0.2 * 0.3, 0.1 * 0.3 + 0.2 * 0.3, 0.05])
    assert np.allclose(c, want, rtol=1e-15, atol=0)
    rare = 0.1 * 0.2 + 0.1 * 0.3 + 0.2 * 0.3 + 0.05
    m = importance_measures(c, p, rare)
    assert np.allclose(m["fussell_vesely"], want / rare) and np.allclose(m["birnbaum"], [0.5, 0.4, 0.3, 1.0])
    assert abs(m["fussell_vesely"].sum() - (2 * (rare - 0.05) + 0.05) / rare) < 1e-15  # each pair counted twice
    # the cut-off removes {ab} (0.02 < 0.025) from every sum; limit_order 1 keeps {d} only
    c2 = oracle.importance(row_ptr, col, p, cut_off=0.025)
    assert np.allclose(c2, [0.03, 0.06, 0.09, 0.05], rtol=1e-15)
    c3 = oracle.importance(row_ptr, col, p, cut_off=0.0, limit_order=1)
    assert np.array_equal(c3 != 0, [False, False, False, True])
    # contributions add up to sum_i len_i * P_i over the kept rows
    rp, cl = oracle.synth_csr(SEED, 5000, 300, 4)
    pp = oracle.synth_probabilities(SEED, 300)
    q = oracle.quantify(rp, cl, pp, cut_off=1e-10)
    ci = oracle.importance(rp, cl, pp, cut_off=1e-10)
    lens = np.diff(rp)
    keep = (q.keep_mask[np.arange(5000) // 64] >> (np.arange(5000) % 64).astype(np.uint64)) & np.uint64(1)
    assert abs(ci.sum() - float((np.exp(q.row_logp) * lens * keep).sum())) <= 1e-13 * ci.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("n_rows,n_events,avg,cut_off", [
    (40_000, 800, 4, 1e-8),        # ROWS_LDS
    (60_000, 5_000, 6, 1e-9),      # WAVE_STREAM
    (1_300_000, 20_000, 6, 1e-8),  # FILTER_REFINE (AUTO from 1e6 rows)
    (30_000, 600, 4, 0.0),         # no cut-off: every row contributes
])
def test_gpu_importance_matches_oracle(gpu_ctx, oracle, n_rows, n_events, avg, cut_off):
    import canopy_amd
    from canopy_amd import _capi
    row_ptr, col = oracle.synth_csr(SEED, n_rows, n_events, avg, n_threads=4)
    p = canopy_amd.synthetic_probabilities(SEED, n_events)
    want = oracle.importance(row_ptr, col, p, cut_off=cut_off)
    model = gpu_ctx.upload(row_ptr, col, n_events)
    model.set_probabilities(p)
    prm = canopy_amd.default_params(cut_off=cut_off, flags=_capi.QUANT_IMPORTANCE)
    res = gpu_ctx.quantify(model, prm)
    got = gpu_ctx.fetch_importance(model)
    assert np.array_equal(got == 0, want == 0)
    nz = want != 0
    assert np.max(np.abs(got[nz] - want[nz]) / want[nz]) < TOL
    plain = gpu_ctx.quantify(model, canopy_amd.default_params(cut_off=cut_off))  # the flag changes nothing else
    assert (plain.rare_event, plain.mcub, plain.n_kept, plain.argmax_row) == (res.rare_event, res.mcub, res.n_kept, res.argmax_row)
    with pytest.raises(_capi.CanopyError) as e:  # ... and the last pass did not ask for importance
        gpu_ctx.fetch_importance(model)
    assert e.value.status == _capi.ERR_STATE
    model.close()


@pytest.mark.gpu
def test_gpu_importance_shards_add_up_in_a_caller_buffer(gpu_ctx, oracle):
    """The multi-GPU route: each shard accumulates into a caller-owned device buffer (the all-reduce operand);
    the shard contributions of a row-partitioned model sum to the whole model's."""
    import torch
    import canopy_amd
    from canopy_amd import _capi
    n_rows, n_events = 80_000, 1500
    row_ptr, col = oracle.synth_csr(SEED, n_rows, n_events, 5, n_threads=2)
    p = canopy_amd.synthetic_probabilities(SEED, n_events)
    want = oracle.importance(row_ptr, col, p)
    bounds = canopy_amd.shard_rows(row_ptr, 3)
    prm = canopy_amd.default_params(flags=_capi.QUANT_IMPORTANCE)
    total = torch.zeros(n_events, dtype=torch.float64, device="cuda")
    buf = torch.zeros(n_events, dtype=torch.float64, device="cuda")
    gpu_ctx.set_importance_buffer(buf.data_ptr(), n_events)
    try:
        for k in range(3):
            rp = row_ptr[bounds[k]:bounds[k + 1] + 1]
            shard = gpu_ctx.upload(rp - rp[0], col[rp[0]:rp[-1]], n_events, row_offset=int(bounds[k]))
            shard.set_probabilities(p)
            gpu_ctx.quantify(shard, prm)
            gpu_ctx.synchronize()
            total += buf  # what the sum all-reduce does across ranks
            shard.close()
    finally:
        gpu_ctx.set_importance_buffer(None, 0)
    got = total.cpu().numpy()
    nz = want != 0
    assert np.array_equal(got == 0, want == 0) and np.max(np.abs(got[nz] - want[nz]) / want[nz]) < TOL

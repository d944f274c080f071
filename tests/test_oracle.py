"""Pin the CPU oracle (parity unpinned by the reference: it ships no golden vector, KAT or
fixture for this path -- reference tests/CMakeLists.txt is 0 bytes). Pinned here by closed-form
known answers, by the committed 50-digit mpmath vectors (tests/golden/), and by self-consistency."""
import hashlib
import json
import math
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "quantify_mpmath.json")
SEED = 20250704


def csr(rows):
    row_ptr = np.zeros(len(rows) + 1, dtype=np.int64)
    for i, r in enumerate(rows):
        row_ptr[i + 1] = row_ptr[i] + len(r)
    col = np.array([c for r in rows for c in r], dtype=np.int32)
    return row_ptr, col


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


# ---- closed-form known answers ------------------------------------------------------

def test_single_cut_set(oracle):
    p = np.array([0.1, 0.2, 0.3])
    q = oracle.quantify(*csr([[0, 1, 2]]), p, cut_off=0.0)
    assert rel(q.rare_event, 0.006) < 1e-15 and rel(q.mcub, 0.006) < 1e-14
    assert q.n_kept == 1 and q.argmax_row == 0
    assert q.row_logp[0] == (math.log(0.1) + math.log(0.2)) + math.log(0.3)  # CSR order, plain adds


def test_disjoint_single_event_cut_sets(oracle):
    p = np.array([1e-3, 2e-3, 3e-3, 4e-3])
    q = oracle.quantify(*csr([[0], [1], [2], [3]]), p, cut_off=0.0)
    assert rel(q.rare_event, p.sum()) < 1e-15
    assert rel(q.mcub, 1.0 - np.prod(1.0 - p)) < 1e-13
    assert q.argmax_row == 3


def test_two_of_three_voting(oracle):
    p = np.full(3, 0.1)
    q = oracle.quantify(*csr([[0, 1], [0, 2], [1, 2]]), p, cut_off=0.0)
    assert rel(q.rare_event, 0.03) < 1e-15
    assert rel(q.mcub, 1.0 - 0.99 ** 3) < 1e-13  # 0.029701
    assert q.argmax_row == 0  # three-way tie -> lowest index


def test_empty_model_and_empty_row(oracle):
    q = oracle.quantify(np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.array([0.5]))
    assert q.rare_event == 0.0 and q.mcub == 0.0 and q.n_kept == 0 and q.argmax_row == -1
    q = oracle.quantify(*csr([[], [0]]), np.array([0.5]), cut_off=0.0)  # empty cut set = TRUE
    assert q.row_logp[0] == 0.0 and q.n_certain == 1 and q.mcub == 1.0 and rel(q.rare_event, 1.5) < 1e-15


def test_certain_and_impossible_events(oracle):
    q = oracle.quantify(*csr([[0, 1], [2]]), np.array([1.0, 1.0, 0.0]), cut_off=0.0)
    assert q.row_logp[0] == 0.0 and q.row_logp[1] == -math.inf
    assert q.n_kept == 2 and q.n_certain == 1 and q.mcub == 1.0 and q.rare_event == 1.0
    assert q.argmax_row == 0
    q = oracle.quantify(*csr([[2]]), np.array([1.0, 1.0, 0.0]), cut_off=1e-8)
    assert q.n_kept == 0 and q.rare_event == 0.0


def test_cut_off_boundary_is_inclusive(oracle):
    # s_i == log(cut_off) exactly -> kept (p == cut_off gives bit-identical logs); just below -> dropped
    cut = 1e-8
    for pv, kept in ((cut, 1), (cut * (1 - 1e-12), 0), (cut * (1 + 1e-12), 1)):
        q = oracle.quantify(*csr([[0]]), np.array([pv]), cut_off=cut)
        assert q.n_kept == kept
    # two-event row whose sum of logs lands exactly on the threshold: log(1e-4)+log(1e-4) vs log(1e-8)
    s = math.log(1e-4) + math.log(1e-4)
    q = oracle.quantify(*csr([[0, 1]]), np.array([1e-4, 1e-4]), cut_off=cut)
    assert q.n_kept == (1 if s >= math.log(cut) else 0)


def test_limit_order(oracle):
    p = np.full(6, 0.5)
    q = oracle.quantify(*csr([[0, 1, 2], [0, 1], [3, 4, 5], [5]]), p, cut_off=0.0, limit_order=2)
    assert q.n_kept == 2 and list(np.nonzero(np.unpackbits(q.keep_mask.view(np.uint8), bitorder="little"))[0]) == [1, 3]
    assert rel(q.rare_event, 0.75) < 1e-15


def test_invalid_probability_rejected(oracle):
    with pytest.raises(ValueError):
        oracle.quantify(*csr([[0]]), np.array([1.5]))
    with pytest.raises(ValueError):
        oracle.quantify(*csr([[0]]), np.array([float("nan")]))


# ---- mpmath vectors --------------------------------------------------------------------------------

def _gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize("idx", range(6))
def test_against_mpmath_golden(oracle, idx):
    g = _gold()
    c = g["cases"][idx]
    row_ptr, col = oracle.synth_csr(g["seed"], c["n_rows"], g["n_events"], c["avg"], first_row=c["first_row"])
    h = hashlib.sha256()
    h.update(row_ptr.tobytes())
    h.update(col.tobytes())
    assert h.hexdigest() == c["csr_sha256"], "C generator differs from the pure-Python restatement"
    p = np.array([float.fromhex(x) for x in g["p_%s_hex" % c["p"]]])
    for threads in (1, 3):
        q = oracle.quantify(row_ptr, col, p, cut_off=c["cut_off"], limit_order=c["limit_order"], n_threads=threads,
                            row_offset=c["first_row"])
        e = c["expected"]
        assert int(q.n_kept) == e["n_kept"] and int(q.argmax_row) == e["argmax_row"]
        # measured agreement is ~1e-15; 1e-13 leaves room for the rounding of logp = log(p)
        assert rel(q.rare_event, float(e["rare"])) < 1e-13
        assert rel(q.mcub, float(e["mcub"])) < 1e-13
        assert abs(q.max_logp - float(e["max_logp"])) < 1e-13


def test_synthetic_probabilities_match_golden_inputs(oracle):
    g = _gold()
    p = oracle.synth_probabilities(g["seed"], g["n_events"])
    want = np.array([float.fromhex(x) for x in g["p_low_hex"]])
    # pow() is glibc's on both sides; allow 1 ulp in case the box's libm differs
    assert np.all(np.abs(p - want) <= np.spacing(want))
    assert p.min() >= 1e-8 and p.max() < 1e-3


# ---- self-consistency --------------------------------------------------------------------------------

def test_product_domain_agrees(oracle):
    row_ptr, col = oracle.synth_csr(SEED, 100000, 1000, 4)
    p = oracle.synth_probabilities(SEED, 1000)
    q = oracle.quantify(row_ptr, col, p, cut_off=0.0, limit_order=0)
    rare, omp = oracle.product_domain(row_ptr, col, p)
    assert rel(q.rare_event, rare) < 1e-13 and rel(q.mcub, omp) < 1e-12
    assert 1.0e-4 < q.rare_event < 2.0e-4  # ~1.5e-4 at config 2


def test_threads_and_shards_agree(oracle):
    row_ptr, col = oracle.synth_csr(SEED, 50000, 1000, 6)
    p = oracle.synth_probabilities(SEED, 1000)
    q1 = oracle.quantify(row_ptr, col, p, cut_off=1e-12)
    for t in (2, 5, 8):
        qt = oracle.quantify(row_ptr, col, p, cut_off=1e-12, n_threads=t)
        assert qt.n_kept == q1.n_kept and qt.argmax_row == q1.argmax_row and qt.max_logp == q1.max_logp
        assert rel(qt.rare_event, q1.rare_event) < 1e-14 and rel(qt.mcub, q1.mcub) < 1e-14
        assert np.array_equal(qt.row_logp, q1.row_logp) and np.array_equal(qt.keep_mask, q1.keep_mask)
    for n in (2, 4, 8):
        b = oracle.shard_rows(row_ptr, n)
        assert b[0] == 0 and b[-1] == 50000 and np.all(np.diff(b) >= 0)
        nnz = np.diff(row_ptr[b])
        assert nnz.max() - nnz.min() <= 2 * 10  # balanced by nnz to within a row or two
        parts = []
        for k in range(n):
            rp = row_ptr[b[k]:b[k + 1] + 1]
            parts.append(oracle.quantify(rp - rp[0], col[rp[0]:rp[-1]], p, cut_off=1e-12, row_offset=int(b[k])).res)
        c = oracle.combine(parts)
        assert c.n_kept == q1.n_kept and c.argmax_row == q1.argmax_row
        assert rel(c.rare_event, q1.rare_event) < 1e-14 and rel(c.mcub, q1.mcub) < 1e-14


def test_row_permutation_invariance(oracle):
    row_ptr, col = oracle.synth_csr(SEED, 5000, 1000, 4)
    p = oracle.synth_probabilities(SEED, 1000)
    q = oracle.quantify(row_ptr, col, p, cut_off=1e-9)
    perm = np.random.default_rng(1).permutation(5000)
    rows = [col[row_ptr[i]:row_ptr[i + 1]] for i in perm]
    q2 = oracle.quantify(*csr(rows), p, cut_off=1e-9)
    assert q2.n_kept == q.n_kept and np.array_equal(q2.row_logp, q.row_logp[perm])
    assert perm[int(q2.argmax_row)] == q.argmax_row or q2.max_logp == q.max_logp
    assert rel(q2.rare_event, q.rare_event) < 1e-14


def test_generator_is_counter_based(oracle):
    a_rp, a_col = oracle.synth_csr(SEED, 3000, 100000, 6)
    b_rp, b_col = oracle.synth_csr(SEED, 1000, 100000, 6, first_row=1500, n_threads=3)
    lo, hi = a_rp[1500], a_rp[2500]
    assert np.array_equal(b_rp, a_rp[1500:2501] - lo) and np.array_equal(b_col, a_col[lo:hi])
    lens = np.diff(a_rp)
    assert lens.min() >= 2 and lens.max() <= 10 and abs(lens.mean() - 6) < 0.2
    for i in range(0, 3000, 97):
        r = a_col[a_rp[i]:a_rp[i + 1]]
        assert np.all(np.diff(r) > 0) and r.min() >= 0 and r.max() < 100000

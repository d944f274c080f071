"""N>1 path on CPU: world_size-2 (and 4) gloo processes. Each rank owns an nnz-balanced row shard,
computes its partial record (here with the oracle -- this is a test of the host logic and the
single-collective combine, the GPU kernels are covered by the -m gpu tests), and the ranks meet
in ONE sum all-reduce through canopy_amd.distributed.ShardCombiner."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20250704


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_rows, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_py
    from canopy_amd import _capi, shard_rows
    from canopy_amd.distributed import ShardCombiner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        row_ptr, col = oracle_py.synth_csr(SEED, n_rows, 1000, 6)
        p = oracle_py.synth_probabilities(SEED, 1000)
        b = shard_rows(row_ptr, world)
        rp = row_ptr[b[rank]:b[rank + 1] + 1]
        q = oracle_py.quantify(rp - rp[0], col[rp[0]:rp[-1]], p, cut_off=1e-12, row_offset=int(b[rank])).res
        comb = ShardCombiner(ctx=None)
        comb.put_local(_capi.Partial(q.rare_hi, q.rare_lo, q.log1p_hi, q.log1p_lo, q.n_kept, q.n_certain, q.max_logp,
                                     q.argmax_row))
        comb.all_reduce()  # the one collective
        r = comb.result()
        np.save(os.path.join(out_dir, "r%d.npy" % rank),
                np.array([r.rare_event, r.mcub, r.n_kept, r.argmax_row, r.max_logp]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])  # 8 = the node size the scaling run uses
def test_sharded_quantification_gloo(tmp_path, oracle, canopy_lib, world):
    n_rows = 40000
    mp.spawn(_worker, args=(world, _free_port(), n_rows, str(tmp_path)), nprocs=world, join=True)
    row_ptr, col = oracle.synth_csr(SEED, n_rows, 1000, 6)
    p = oracle.synth_probabilities(SEED, 1000)
    whole = oracle.quantify(row_ptr, col, p, cut_off=1e-12)
    res = [np.load(os.path.join(str(tmp_path), "r%d.npy" % r)) for r in range(world)]
    for r in res[1:]:
        assert np.array_equal(r, res[0])  # every rank holds the bitwise-identical result
    rare, mcub, n_kept, argmax, max_logp = res[0]
    assert n_kept == whole.n_kept and argmax == whole.argmax_row and max_logp == whole.max_logp
    assert abs(rare - whole.rare_event) <= 1e-12 * whole.rare_event
    assert abs(mcub - whole.mcub) <= 1e-12 * whole.mcub


def _batch_worker(rank, world, port, n_rows, n_trials, out_dir):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import oracle_py
    from canopy_amd import _capi, shard_rows
    from canopy_amd.distributed import BatchCombiner
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        row_ptr, col = oracle_py.synth_csr(SEED, n_rows, 1000, 6)
        pt = _trials(oracle_py.synth_probabilities(SEED, 1000), n_trials)
        b = shard_rows(row_ptr, world)
        rp = row_ptr[b[rank]:b[rank + 1] + 1]
        parts = []
        for k in range(n_trials):
            q = oracle_py.quantify(rp - rp[0], col[rp[0]:rp[-1]], pt[k], cut_off=1e-10, row_offset=int(b[rank]),
                                   want_rows=False).res
            parts.append(_capi.Partial(q.rare_hi, q.rare_lo, q.log1p_hi, q.log1p_lo, q.n_kept, q.n_certain, q.max_logp,
                                       q.argmax_row))
        comb = BatchCombiner(n_trials)
        comb.put_local(parts)
        comb.all_gather()  # the one collective of the batch
        res = comb.results()
        np.save(os.path.join(out_dir, "b%d.npy" % rank),
                np.array([[r.rare_event, r.mcub, r.n_kept, r.argmax_row, r.max_logp] for r in res]))
    finally:
        dist.destroy_process_group()


def _trials(p, n_trials):
    rng = np.random.default_rng(3)
    return np.minimum(1.0, p[None, :] * np.exp(0.5 * rng.standard_normal((n_trials, p.size))))


def test_sharded_batch_gloo(tmp_path, oracle, canopy_lib):
    world, n_rows, n_trials = 2, 20000, 5
    mp.spawn(_batch_worker, args=(world, _free_port(), n_rows, n_trials, str(tmp_path)), nprocs=world, join=True)
    row_ptr, col = oracle.synth_csr(SEED, n_rows, 1000, 6)
    pt = _trials(oracle.synth_probabilities(SEED, 1000), n_trials)
    res = [np.load(os.path.join(str(tmp_path), "b%d.npy" % r)) for r in range(world)]
    assert np.array_equal(res[0], res[1])
    for k in range(n_trials):
        whole = oracle.quantify(row_ptr, col, pt[k], cut_off=1e-10, want_rows=False)
        rare, mcub, n_kept, argmax, max_logp = res[0][k]
        assert n_kept == whole.n_kept and argmax == whole.argmax_row and max_logp == whole.max_logp
        assert abs(rare - whole.rare_event) <= 1e-12 * whole.rare_event
        assert abs(mcub - whole.mcub) <= 1e-12 * whole.mcub


def test_bench_launcher_starts_its_ranks_and_reports_a_dead_rank():
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): the parent spawns its ranks before touching torch or the GPU
    and relays their failure as a non-zero exit code instead of hanging. On this CPU-only container every rank stops with
    "needs an MI355X" -- which is exactly the path "a rank died" of the launcher; on the GPU box the same command runs the
    benchmark (profiles/r03_bench_n2_launcher_rehearsal.json)."""
    import subprocess
    import time
    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the CPU-side check of the launcher; the GPU run is benchmarks/evidence.sh")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--watchdog-s", "120"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert r.stdout.strip() == ""  # no JSON line from a run that did not measure anything
    assert "needs an MI355X" in r.stderr and "stopping the others" in r.stderr or "needs an MI355X" in r.stderr
    assert time.time() - t0 < 200

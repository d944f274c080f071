"""canopy_comm at world size 2 and 3 on ONE GPU (-m gpu): rank processes with a TEST DOUBLE of RCCL preloaded
(tests/rccl_shim/rccl_shim.cpp: asynchronous stream work, the k-th collective of every rank meets the k-th of the others, one
host-function thread per process -- two collectives of a communicator that become runnable out of issue order dead-lock).
Real RCCL refuses two ranks on one device and the project's GPU box has one: this is what exercises, with real concurrency,
the per-lane slot buffers [world][8], the chaining of consecutive collectives across lanes, the ring of published records
and the ticketed fetch at world > 1. Every rank's combined record of every pass against the oracle's whole-model result."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SEED = 20250704
TOL = 1e-12


def perturbed(p, k):
    rng = np.random.default_rng(4000 + k)
    return np.clip(p * np.exp(0.3 * rng.standard_normal(p.shape)), 0.0, 1.0)


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    out = tmp_path_factory.mktemp("rccl_shim")
    lib = os.path.join(str(out), "librccl.so.1")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wl,-soname,librccl.so.1", "-o", lib,
                    os.path.join(HERE, "rccl_shim", "rccl_shim.cpp"), "-lrt"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return lib


@pytest.mark.parametrize("world", [2, 3])
def test_canopy_comm_world_gt_1_on_one_gpu(canopy_lib, oracle, shim, tmp_path, world):
    total_rows, events = 3_000_000, 20_000
    idfile = str(tmp_path / "comm_id")
    env = dict(os.environ)
    env["LD_PRELOAD"] = shim + (":" + env["LD_PRELOAD"] if env.get("LD_PRELOAD") else "")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_shim", "rank_main.py"), str(r), str(world), idfile, str(tmp_path),
                               str(total_rows), str(events)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(world)]
    try:
        outs = [p.communicate(timeout=400) for p in procs]
    finally:
        for p in procs:  # exactly the processes started above
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, se.decode()[-3000:])
    res = [json.load(open(str(tmp_path / ("rank%d.json" % r)))) for r in range(world)]
    # K6: the sample range split over the ranks, hits summed by the collective = the oracle over the whole range
    dag = {"n_events": 4, "gate_type": np.array([1, 0], dtype=np.int32), "gate_min": np.array([0, 0], dtype=np.int32),
           "arg_ptr": np.array([0, 2, 4], dtype=np.int32), "args": np.array([0, 1, 4, 2], dtype=np.int32), "top": 5}
    want_hits = oracle.mc(dag, np.array([0.3, 0.2, 0.5, 0.9]), 11, 64 * 30_000, n_threads=2)
    assert all(res[r]["mc"] == [want_hits, 64 * 30_000] for r in range(world)), [res[r]["mc"] for r in range(world)]
    # expectation: the oracle on the WHOLE model, vector by vector in the order the ranks used them
    row_ptr, col = oracle.synth_csr(SEED, total_rows, events, 6, n_threads=4)
    base = oracle.synth_probabilities(SEED, events)
    step = 0
    for mode in ("chained", "own_stream"):
        for cut_off in (1e-9, 0.0):
            key = "%s/%g" % (mode, cut_off)
            assert all(res[r][key] == res[0][key] for r in range(world)), "ranks disagree on the combined records (%s)" % key
            assert len(res[0][key]) == 13
            for rec in res[0][key]:
                want = oracle.quantify(row_ptr, col, perturbed(base, step), cut_off=cut_off, n_threads=4, want_rows=False)
                step += 1
                assert rec[2] == int(want.n_kept) and rec[3] == int(want.argmax_row) and rec[4] == want.max_logp, key
                assert abs(rec[0] - want.rare_event) <= TOL * want.rare_event and abs(rec[1] - want.mcub) <= TOL * want.mcub, key

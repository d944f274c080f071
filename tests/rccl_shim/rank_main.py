"""One rank of tests/test_comm_world2.py: its shard of a 3-way / 2-way split through canopy_comm at world size > 1 on ONE GPU,
with tests/rccl_shim/rccl_shim.cpp preloaded in place of RCCL. Imports neither torch nor anything else that carries an RCCL.
usage: rank_main.py <rank> <world> <id file> <out dir> <total rows> <events> [modes: chained,own_stream,unordered]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import canopy_amd  # noqa: E402
from canopy_amd import _capi  # noqa: E402

SEED = 20250704


def perturbed(p, k):
    rng = np.random.default_rng(4000 + k)
    return np.clip(p * np.exp(0.3 * rng.standard_normal(p.shape)), 0.0, 1.0)


def main():
    rank, world, idfile, outdir, total_rows, events = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    ctx = canopy_amd.Context(device=0)
    if rank == 0:
        uid = canopy_amd.Comm.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id after 60 s")
            time.sleep(0.01)
        uid = open(idfile, "rb").read()
    comm = canopy_amd.Comm(ctx, uid, world, rank)  # collective
    assert comm.rccl_version() == 99999, "the test double is not what libcanopy bound"
    bounds = canopy_amd.synthetic_shard_bounds(ctx, SEED, total_rows, 6, world)
    model = ctx.synthetic(SEED, int(bounds[rank + 1] - bounds[rank]), events, 6, first_row=int(bounds[rank]))
    base = canopy_amd.synthetic_probabilities(SEED, events)
    out = {}
    step = 0
    modes = {"chained": _capi.COMM_CHAINED, "own_stream": _capi.COMM_OWN_STREAM, "unordered": _capi.COMM_UNORDERED}
    names = sys.argv[7].split(",") if len(sys.argv) > 7 else ["chained", "own_stream"]
    for mode_name, mode in ((n, modes[n]) for n in names):
        comm.set_mode(mode)
        for cut_off in (1e-9, 0.0):
            prm = canopy_amd.default_params(cut_off=cut_off)
            # serial: one pass at a time, its combined result fetched by ticket
            ctx.set_pipeline(0)
            recs = []
            for _ in range(3):
                model.set_probabilities(perturbed(base, step)); step += 1
                r = comm.ticket_fetch(ctx.enqueue(model, prm))
                recs.append([r.rare_event, r.mcub, r.n_kept, r.argmax_row, r.max_logp])
            # pipelined: three lanes, a new vector per pass, up to three passes in flight
            ctx.set_pipeline(3)
            inflight = []
            for _ in range(10):
                model.set_probabilities(perturbed(base, step)); step += 1
                inflight.append(ctx.enqueue(model, prm))
                if len(inflight) == 3:
                    r = comm.ticket_fetch(inflight.pop(0))
                    recs.append([r.rare_event, r.mcub, r.n_kept, r.argmax_row, r.max_logp])
            while inflight:
                r = comm.ticket_fetch(inflight.pop(0))
                recs.append([r.rare_event, r.mcub, r.n_kept, r.argmax_row, r.max_logp])
            last = comm.fetch()  # the legacy fetch: the last pass again
            assert [last.rare_event, last.mcub, last.n_kept, last.argmax_row, last.max_logp] == recs[-1]
            out["%s/%g" % (mode_name, cut_off)] = recs
    # K6: this rank's share of the sample range, one u64 all-reduce of {hits, trials} behind the kernel
    from canopy_amd.quantifier import Dag
    dag = {"n_events": 4, "gate_type": [1, 0], "gate_min": [0, 0], "arg_ptr": [0, 2, 4], "args": [0, 1, 4, 2], "top": 5}
    d = Dag(ctx, dag)
    d.set_probabilities(np.array([0.3, 0.2, 0.5, 0.9]))
    n_total = 64 * 30_000
    per = n_total // world // 64 * 64
    first = rank * per
    mine = n_total - first if rank == world - 1 else per
    d.run_async(11, mine, first_sample=first)
    both = comm.mc_fetch()
    out["mc"] = [int(both.hits), int(both.num_trials)]
    d.close()
    out["variant"] = ctx.last_kernel_variant()
    with open(os.path.join(outdir, "rank%d.json" % rank), "w") as f:
        json.dump(out, f)
    comm.close()
    model.close()
    ctx.close()


if __name__ == "__main__":
    main()

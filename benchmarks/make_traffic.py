#!/usr/bin/env python3
"""Turn the PMC passes of benchmarks/pmc.sh into profiles/traffic*.json: HBM bytes per launch of a two-kernel pass
(FILTER_REFINE: k_fr_filter_ls / k_fr_filter_q8o + k_fr_refine<false, ..>; FIXED_SUM: k_fx2 + k_fr_refine<true, ..>), corrected as
MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE (KiB) x 1024 x 2 (gfx950 tallies the 128-byte requests of a
wide coalesced stream at 64 bytes), WRITE_SIZE (KiB) x 1024 as read.
usage: make_traffic.py <pmc outdir> <out json> [rows events cut_off variant round]"""
import collections
import csv
import glob
import json
import re
import sys


def main():
    out_dir, out_json = sys.argv[1], sys.argv[2]
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000_000
    events = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
    cut_off = float(sys.argv[5]) if len(sys.argv) > 5 else 1e-8
    variant = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    rnd = int(sys.argv[7]) if len(sys.argv) > 7 else 2
    kernels = ("k_fr_filter_", "k_fr_refine<false") if variant == 5 else ("k_fx2", "k_fr_refine<true")  # (k_fr_filter_ls<17>, k_fr_refine<false, true> with the row stream)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    seen = {}  # the kernels' names as the profiler reports them (k_fr_filter_q8o<4> ...)
    for f in glob.glob(out_dir + "/p*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            key = next((k for k in kernels if k in name), None)
            if key:
                m = re.search(r"(k_f[rx][a-z0-9_]*(<[^>]*>)?)", name)
                seen[key] = m.group(1) if m else name
            if key and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT", "TCC_MISS"):
                agg[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
    mean = lambda k, c: (sum(agg[k][c]) / len(agg[k][c])) if agg[k][c] else 0.0
    per_kernel = {}
    total = 0.0
    for k in kernels:
        rd = mean(k, "FETCH_SIZE") * 1024 * 2
        wr = mean(k, "WRITE_SIZE") * 1024
        per_kernel[k] = {"FETCH_SIZE_KB": mean(k, "FETCH_SIZE"), "WRITE_SIZE_KB": mean(k, "WRITE_SIZE"), "read_bytes": rd,
                         "write_bytes": wr, "TCC_HIT": mean(k, "TCC_HIT"), "TCC_MISS": mean(k, "TCC_MISS"),
                         "launches_sampled": len(agg[k]["FETCH_SIZE"])}
        total += rd + wr
    json.dump({
        "round": rnd, "kernel": " + ".join(seen.get(k, k) for k in kernels), "variant": variant, "rows": rows, "n_events": events, "cut_off": cut_off,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE TCC_HIT TCC_MISS, separate passes over bench.py --steps 3 "
                  "--warmup 1 (benchmarks/pmc.sh)",
        "correction": "gfx950: read bytes = FETCH_SIZE*1024*2 (MI355X_MICROARCH.md, HBM), WRITE_SIZE*1024 exact",
        "note": "the x2 is calibrated for wide coalesced streams (the stream of the row kernel); for the exact pass's narrow "
                "random reads, and for the table gathers of k_fx2, the guide calls the counter uncalibrated: their share is an "
                "upper bound",
        "per_kernel": per_kernel, "hbm_bytes_per_launch": total,
    }, open(out_json, "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_launch": total, "per_kernel": per_kernel}))


if __name__ == "__main__":
    main()

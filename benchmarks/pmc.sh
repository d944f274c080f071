#!/bin/bash
# PMC passes over bench.py (one counter group per run; no tracing flags -- gpurun refuses the mix).
# usage: benchmarks/pmc.sh <outdir> [bench args...]     (each pass is bounded by `timeout`)
# PMC_SCRIPT=benchmarks/bench_mc.py PMC_ARGS="--steps 1" profiles another driver (default: bench.py, 3 steps)
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
  "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VALU" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL" \
  "FETCH_SIZE" \
  "WRITE_SIZE TCC_HIT TCC_MISS" ; do
  i=$((i+1))
  echo "pass $i: $grp" >> "$OUT/progress.log"
  timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d "$OUT/p$i" -o pmc -- python3 "$GRAFT_REPO_ROOT/${PMC_SCRIPT:-bench.py}" ${PMC_ARGS:---steps 3 --warmup 1 --no-cpu-baseline} "$@" > "$OUT/p$i.log" 2>&1 || echo "pass $i failed rc=$?" >> "$OUT/progress.log"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        m = re.search(r"(k_fr_filter_q8|k_fr_filter|k_fr_refine|k_fx2|k_fx_sum|k_ws_build|k_fr_build|k_bs_place|k_bs_sort|k_bs_alloc|k_bs|k_ws|k_rows|k_vs|k_finalize|k_mc|k_batch_rows)", k)
        if m and m.group(1) in ("k_bs", "k_ws", "k_rows", "k_vs", "k_fr_filter_q8", "k_fr_filter", "k_fr_refine", "k_fx2", "k_fx_sum", "k_mc", "k_batch_rows"):
            agg[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in agg.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print("  %-36s n=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY

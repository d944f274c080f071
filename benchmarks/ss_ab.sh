set -u
cd $GRAFT_REPO_ROOT
export CANOPY_AMD_LIB=$PWD/benchmarks/bin/libcanopy_A.so
for rep in 1 2 3; do
  for ss in 4 0; do
    for sh in 8 1; do
      if [ $sh = 1 ]; then a=""; else a="--shard-of $sh --force-dist"; fi
      ms=$(CANOPY_LS_SS=$ss timeout -k 10 200 python bench.py $a --no-also --no-cpu-baseline --steps 400 --warmup 40 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['serial']['ms_per_step'], d['roofline']['median_kernel_ms'])")
      echo "rep $rep forced_ss $ss shard 1/$sh: $ms"
    done
  done
done

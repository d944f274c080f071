set -u
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in A B; do
  CANOPY_AMD_LIB=$PWD/benchmarks/bin/libcanopy_$v.so timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); a=d['also']
print('$v $rep c0 step %.4f kernel %.4f | row_logp kernel %.4f' % (a['cut_off_0']['ms_per_step'], a['cut_off_0']['kernel_ms']['median'], a['row_logp']['kernel_ms']['median']))"
done; done

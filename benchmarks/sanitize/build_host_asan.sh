# Host code under ASan + UBSan (device code untouched): builds $OUT/libcanopy_asan.so; then
#   LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \
#   CANOPY_AMD_LIB=$OUT/libcanopy_asan.so python -m pytest tests -q -m "not gpu" --deselect tests/test_abi.py::test_default_library_carries_no_tuning_switches
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${OUT:-/tmp/asan_build}; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -fvisibility=hidden -DCANOPY_BUILD -DCANOPY_SHARED -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer -I$R/include -I$R/canopy_amd/csrc"
pids=""
for f in canopy_api.hip quantify_pass.hip host_pool.cpp kernels_rows.hip kernels_build.hip kernels_ws.hip kernels_bs.hip kernels_ls.hip kernels_vs.hip kernels_mc.hip kernels_fr.hip kernels_batch.hip comm_rccl.hip mef_lowering.cpp; do
  /opt/rocm/bin/hipcc $FLAGS -c $R/canopy_amd/csrc/$f -o $OUT/$f.o &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -fsanitize=address -fsanitize=undefined -shared-libsan $OUT/*.o -ldl -o $OUT/libcanopy_asan.so
echo built

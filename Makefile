# Convenience targets; the build itself is canopy_amd/build.py (explicit hipcc, in-tree libcanopy.so).
.PHONY: build oracle test-cpu test-gpu bench bench-cpp evidence clean
build:
	python -m canopy_amd.build
oracle:
	$(MAKE) -C oracle
test-cpu: build oracle
	python -m pytest tests -x -q -m "not gpu"
test-gpu: build oracle
	python -m pytest tests -x -q -m gpu
bench: build
	python bench.py
bench-cpp: build
	/opt/rocm/bin/hipcc -std=c++17 -O2 -Iinclude benchmarks/bench_quantify.cpp -Lcanopy_amd -lcanopy -Wl,-rpath,$(CURDIR)/canopy_amd -o benchmarks/bench_quantify
	./benchmarks/bench_quantify
evidence: build
	bash benchmarks/evidence.sh r01
clean:
	rm -rf canopy_amd/build canopy_amd/libcanopy.so oracle/libcanopy_oracle.so benchmarks/bench_quantify examples/quantify_mef

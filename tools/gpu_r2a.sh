#!/bin/bash
# round-2 GPU session A: correctness of the refactored library + new plane-split x3 kernel, layer timings, bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r2a_pytest.log 2>&1
( BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so timeout 300 python tools/bench_layers.py --nterms 3 --classes 0,4 2>&1 | tail -30 ) > gpurun_out/r2a_layers_x3.log 2>&1
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -5 ) > gpurun_out/r2a_bench.log 2>&1
( timeout 300 python bench.py --no-cpu-baseline --no-extras --batched 2>&1 | tail -3 ) > gpurun_out/r2a_bench_batched.log 2>&1
tail -5 gpurun_out/r2a_pytest.log; cat gpurun_out/r2a_layers_x3.log; cat gpurun_out/r2a_bench.log; cat gpurun_out/r2a_bench_batched.log

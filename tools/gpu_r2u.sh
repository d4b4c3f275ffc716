#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( WGRAD=1 BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so timeout 300 python tools/bench_layers.py --nterms 3 --classes 9 2>&1 | grep wgrad ) > gpurun_out/r2u_wgrad.log 2>&1
cat gpurun_out/r2u_wgrad.log
( timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -3 )
( timeout 300 python bench.py --no-cpu-baseline --no-extras --mode train --steps 8 2>&1 | tail -1 | cut -c1-200 )

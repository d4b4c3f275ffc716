#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round2.py -q -x 2>&1 | grep -E "passed|failed|rror|^E " | tail -8 ) 2>&1 | tee gpurun_out/wab_pytest.log
BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so WG_DBGS=0,1,2,64,32 timeout 300 python tools/bench_wgrad.py 2>&1 | tail -5 | tee gpurun_out/wab_bench.log
( timeout 300 python bench.py --mode train 2>&1 | tail -1 ) > gpurun_out/wab_bench_train.json 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/wab_bench_train.json

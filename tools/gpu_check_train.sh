#!/bin/bash
# backward-path check: the backward / round-2 GPU tests, the training bench line, and the training kernel table
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round2.py -q -x 2>&1 | grep -E "passed|failed|rror|^E " | tail -8 ) > gpurun_out/ct_pytest.log 2>&1
cat gpurun_out/ct_pytest.log
( timeout 300 python bench.py --mode train 2>&1 | tail -1 ) > gpurun_out/ct_bench_train.json 2>&1
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ct_bench_train.json
rm -rf gpurun_out/prof_ct
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ct -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 > gpurun_out/prof_ct.log 2>&1
python tools/stats_md.py gpurun_out/prof_ct 12 > gpurun_out/ct_stats_train.md 2>&1
cat gpurun_out/ct_stats_train.md

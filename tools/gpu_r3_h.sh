#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py -q -x 2>&1 | grep -E "passed|failed|rror|^E |^FAILED" | tail -5 )
rm -rf gpurun_out/prof_r3h
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3h/kt_train -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 > /dev/null 2>&1
python tools/stats_md.py gpurun_out/prof_r3h/kt_train 16
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3h/kt_x3 -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python tools/stats_md.py gpurun_out/prof_r3h/kt_x3 9
find gpurun_out/prof_r3h -name "*kernel_trace.csv" -delete
for rep in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '
timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --precision f16 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r2/kt_train4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_train4 -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 > gpurun_out/prof_r2_kt_train4.log 2>&1
python tools/stats_md.py gpurun_out/prof_r2/kt_train4 26 > gpurun_out/r2o_stats_train4.md
find gpurun_out/prof_r2 -name "*kernel_trace.csv" -delete
cat gpurun_out/r2o_stats_train4.md

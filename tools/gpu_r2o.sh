#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_r2/kt_train4
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_train4 -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 > gpurun_out/prof_r2_kt_train4.log 2>&1
python tools/stats_md.py gpurun_out/prof_r2/kt_train4 20 > gpurun_out/r2o_stats_train4_serial.md
find gpurun_out/prof_r2 -name "*kernel_trace.csv" -delete
cat gpurun_out/r2o_stats_train4_serial.md
tools/smi_watch.sh gpurun_out/r2o_smi_train.log -- timeout 300 python bench.py --mode train --steps 12 > gpurun_out/r2o_bench_train.log 2>&1
grep -o '"value": [0-9.]*, "unit": "[a-z /]*", "n_gpus": 1, "steps": [0-9]*[^}]*"ms_per_step": [0-9.]*' gpurun_out/r2o_bench_train.log | head -1
sed 's/GPU\[0\]\t\t: //g' gpurun_out/r2o_smi_train.log | awk '{print $5, $NF}' | sort | uniq -c | sort -rn | head -12

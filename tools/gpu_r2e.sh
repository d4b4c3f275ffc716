#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocm-smi --showclocks --showpower 2>&1 | head -20 > gpurun_out/r2e_smi_idle.log
tools/smi_watch.sh gpurun_out/r2e_smi_x3.log -- timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 > gpurun_out/r2e_bench_x3.log 2>&1
tools/smi_watch.sh gpurun_out/r2e_smi_x3_zero.log -- timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --zero-data > gpurun_out/r2e_bench_x3_zero.log 2>&1
tools/smi_watch.sh gpurun_out/r2e_smi_f16.log -- timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --precision f16 > gpurun_out/r2e_bench_f16.log 2>&1
tools/smi_watch.sh gpurun_out/r2e_smi_f16_zero.log -- timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 40 --precision f16 --zero-data > gpurun_out/r2e_bench_f16_zero.log 2>&1
cat gpurun_out/r2e_smi_idle.log
for t in x3 x3_zero f16 f16_zero; do
  echo "== $t"; grep -o '"value": [0-9.]*, "unit": "interpolated frames/s", "n_gpus": 1, "steps": 40[^}]*"ms_per_step": [0-9.]*' gpurun_out/r2e_bench_$t.log | head -1
  grep -o '"avg_kernel_us": [0-9.]*' gpurun_out/r2e_bench_$t.log | head -1
  sort gpurun_out/r2e_smi_$t.log | uniq -c | sort -rn | head -6
done

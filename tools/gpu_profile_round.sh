#!/bin/bash
# final round-2 refresh: tests, kernel stats, per-precision PMC traffic, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -4 ) > gpurun_out/r2v_pytest.log 2>&1
cat gpurun_out/r2v_pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf gpurun_out/prof_r2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_x3 -o kt -- $B > gpurun_out/prof_r2_kt_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_f16 -o kt -- $B --precision f16 > gpurun_out/prof_r2_kt_f16.log 2>&1
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_train -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 > gpurun_out/prof_r2_kt_train.log 2>&1
python tools/stats_md.py gpurun_out/prof_r2/kt_x3 16 > gpurun_out/r2v_stats_x3.md
python tools/stats_md.py gpurun_out/prof_r2/kt_f16 12 > gpurun_out/r2v_stats_f16.md
python tools/stats_md.py gpurun_out/prof_r2/kt_train 22 > gpurun_out/r2v_stats_train.md
P="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --calib"
rm -f gpurun_out/r2v_pmc_traffic.json
for prec in f16x3 f16; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$prec -- $P --precision $prec > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$prec -- $P --precision $prec > /dev/null 2>&1
  python tools/pmc_traffic.py /tmp/pmc_f_$prec /tmp/pmc_w_$prec --json gpurun_out/r2v_pmc_traffic.json --key $prec > gpurun_out/r2v_pmc_traffic_$prec.md
done
find gpurun_out/prof_r2 -name "*kernel_trace.csv" -delete
( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r2v_bench.json 2>&1
( timeout 300 python bench.py --mode train 2>&1 | tail -1 ) > gpurun_out/r2v_bench_train.json 2>&1
head -12 gpurun_out/r2v_stats_x3.md; head -12 gpurun_out/r2v_pmc_traffic_f16x3.md
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2v_bench.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], d["roofline"]["traffic"], d["tolerance_mode"]["value"], d["streaming"]["value"], d["train"].get("ms_per_step"), d["cpu_baseline"]["value"])
d = json.loads(open("gpurun_out/r2v_bench_train.json").read())
print(d["value"], d["ms_per_step"])
PY

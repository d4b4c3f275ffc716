#!/bin/bash
# final round-3 refresh: tests, kernel stats (rocprofv3 --kernel-trace --stats), per-precision PMC traffic, bench lines.
# usage: gpurun -- bash tools/gpu_profile_r3.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
T=gpurun_out/r3v
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -4 ) > ${T}_pytest.log 2>&1
cat ${T}_pytest.log
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf gpurun_out/prof_r3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3/kt_x3 -o kt -- $B > ${T}_kt_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3/kt_f16 -o kt -- $B --precision f16 > ${T}_kt_f16.log 2>&1
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r3/kt_train -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 > ${T}_kt_train.log 2>&1
python tools/stats_md.py gpurun_out/prof_r3/kt_x3 16 > ${T}_stats_x3.md
python tools/stats_md.py gpurun_out/prof_r3/kt_f16 12 > ${T}_stats_f16.md
python tools/stats_md.py gpurun_out/prof_r3/kt_train 24 > ${T}_stats_train.md
tail -1 ${T}_kt_x3.log > ${T}_kt_x3_bench.json
P="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --calib"
rm -f ${T}_pmc_traffic.json
for prec in f16x3 f16; do
  rm -rf /tmp/pmc_f_$prec /tmp/pmc_w_$prec
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$prec -- $P --precision $prec > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$prec -- $P --precision $prec > /dev/null 2>&1
  python tools/pmc_traffic.py /tmp/pmc_f_$prec /tmp/pmc_w_$prec --json ${T}_pmc_traffic.json --key $prec > ${T}_pmc_traffic_$prec.md
done
PT="python bench.py --mode train --batch 8 --steps 1 --warmup 1"
rm -rf /tmp/pmc_tf /tmp/pmc_tw
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tf -- $PT > /dev/null 2>&1
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tw -- $PT > /dev/null 2>&1
python tools/pmc_traffic.py /tmp/pmc_tf /tmp/pmc_tw --json ${T}_pmc_traffic.json --key wgrad3x3 > ${T}_pmc_traffic_train.md
find gpurun_out/prof_r3 -name "*kernel_trace.csv" -delete
cp ${T}_pmc_traffic.json profiles/r03_pmc_traffic.json   # the bench lines below quote THIS run's PMC traffic (box-local copy)
( timeout 900 python bench.py --steps 20 --warmup 5 2>&1 | tail -1 ) > ${T}_bench.json 2>&1
( timeout 300 python bench.py --mode train 2>&1 | tail -1 ) > ${T}_bench_train.json 2>&1
head -12 ${T}_stats_x3.md; head -8 ${T}_stats_train.md; head -8 ${T}_pmc_traffic_f16x3.md; head -8 ${T}_pmc_traffic_train.md
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3v_bench.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], d["roofline"]["traffic"], d["power"], d["tolerance_mode"]["value"], d["streaming"]["value"], d["train"].get("ms_per_step"), d["cpu_baseline"]["value"])
d = json.loads(open("gpurun_out/r3v_bench_train.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["dominant_kernel"]["avg_kernel_us"])
PY

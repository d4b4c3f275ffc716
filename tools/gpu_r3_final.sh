#!/bin/bash
# last call of the round: smoke(), then the two bench lines exactly as the driver runs them (default flags)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 ) | tee gpurun_out/r3v_smoke.log
( time timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r3v_bench.json 2> gpurun_out/r3v_bench_time.log
( timeout 300 python bench.py --mode train 2>&1 | tail -1 ) > gpurun_out/r3v_bench_train.json 2>&1
cat gpurun_out/r3v_bench_time.log
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r3v_bench.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], d["roofline"]["traffic"], d["power"]["clock_mhz"], d["tolerance_mode"]["value"], d["streaming"]["value"], d["train"].get("ms_per_step"), d["cpu_baseline"]["value"])
d = json.loads(open("gpurun_out/r3v_bench_train.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["dominant_kernel"]["avg_kernel_us"], d["roofline"]["dominant_kernel"]["traffic"])
PY

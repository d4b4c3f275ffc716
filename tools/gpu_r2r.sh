#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -5 ) > gpurun_out/r2r_pytest.log 2>&1
cat gpurun_out/r2r_pytest.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke ) > gpurun_out/r2r_smoke.log 2>&1
cat gpurun_out/r2r_smoke.log
( timeout 900 python bench.py 2>&1 | tail -1 ) > gpurun_out/r2r_bench.json 2>&1
( timeout 300 python bench.py --mode train 2>&1 | tail -1 ) > gpurun_out/r2r_bench_train.json 2>&1
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2r_bench.json").read())
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_kernel_us"], d["roofline"]["traffic"], d["tolerance_mode"]["value"], d["streaming"]["value"], d["train"].get("ms_per_step"), d["cpu_baseline"])
d = json.loads(open("gpurun_out/r2r_bench_train.json").read())
print(d["value"], d["ms_per_step"])
PY

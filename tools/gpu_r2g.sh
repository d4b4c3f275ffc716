#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1200 python -m pytest tests/test_gpu_round2.py -m gpu -x -q 2>&1 | tail -30 ) > gpurun_out/r2g_pytest_new.log 2>&1
tail -30 gpurun_out/r2g_pytest_new.log
( timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_gpu_round2.py 2>&1 | tail -12 ) > gpurun_out/r2g_pytest_all.log 2>&1
tail -6 gpurun_out/r2g_pytest_all.log
( timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 ) > gpurun_out/r2g_bench.log 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r2g_bench.log"):
    if ln.startswith("{"):
        d = json.loads(ln)
        print(d["value"], d["ms_per_step"], d["config"]["streams"], d["roofline"]["avg_kernel_us"], d["roofline"]["frac"], d["tolerance_mode"]["value"], d["streaming"], d["train"]["ms_per_step"] if d["train"] and "ms_per_step" in d["train"] else d["train"])
    else:
        print(ln[:300])
PY

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round2.py tests/test_gpu_scripts.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 ) > gpurun_out/r2m_pytest.log 2>&1
cat gpurun_out/r2m_pytest.log
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 ) > gpurun_out/r2m_$tag.log 2>&1; }
for rep in 1 2; do
run train_batched_$rep --mode train --steps 8

done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2m_train*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:45s} {d['value']:8.3f} {d['ms_per_step']:8.2f} ms loss {d['loss']}")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

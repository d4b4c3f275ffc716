#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round2.py tests/test_gpu_scripts.py tests/test_gpu_net.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8 ) > gpurun_out/r2n_pytest.log 2>&1
cat gpurun_out/r2n_pytest.log
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 ) > gpurun_out/r2n_$tag.log 2>&1; }
for rep in 1 2; do
BIN_AMD_FOUR_CALLS=0 run train_17calls_$rep --mode train --steps 8
BIN_AMD_FOUR_CALLS=1 run train_4calls_$rep --mode train --steps 8
BIN_AMD_FOUR_CALLS=1 BIN_AMD_WGRAD_STREAM=0 run train_4calls_nostream_$rep --mode train --steps 8
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2n_train*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:50s} {d['value']:8.3f} {d['ms_per_step']:8.2f} ms loss {d['loss']} mem {d['peak_mem_GB']}")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

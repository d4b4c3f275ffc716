#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|rror" | tail -4 ) > gpurun_out/r2w_pytest.log 2>&1
cat gpurun_out/r2w_pytest.log
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 ) > gpurun_out/r2w_$tag.log 2>&1; }
for rep in 1 2; do
run x3_$rep --steps 30
run f16_$rep --steps 30 --precision f16
run train_$rep --mode train --steps 8
done
rm -rf gpurun_out/prof_r2/kt_x3c
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_x3c -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1
python tools/stats_md.py gpurun_out/prof_r2/kt_x3c 8
find gpurun_out/prof_r2 -name "*kernel_trace.csv" -delete
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2w_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:40s} {d['value']:8.3f} {d['ms_per_step']:8.2f} ms")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

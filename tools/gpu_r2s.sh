#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/check_rdb3.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r2s_check.log 2>&1
cat gpurun_out/r2s_check.log
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 "$@" 2>&1 | tail -1 ) > gpurun_out/r2s_$tag.log 2>&1; }
for rep in 1 2; do
BIN_AMD_RDB3=0 run x3_base_$rep
BIN_AMD_RDB3=1 run x3_rdb3_$rep
done
BIN_AMD_RDB3=0 run train_base --mode train --steps 8
BIN_AMD_RDB3=1 run train_rdb3 --mode train --steps 8
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2s_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:40s} {d['value']:8.3f} {d['ms_per_step']:8.2f} ms")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

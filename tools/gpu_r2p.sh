#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 600 python tools/bench_small.py 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r2p_small.log 2>&1
cat gpurun_out/r2p_small.log
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 "$@" 2>&1 | tail -1 ) > gpurun_out/r2p_$tag.log 2>&1; }
BIN_AMD_INFER_FOUR=0 run x3_17 
BIN_AMD_INFER_FOUR=1 run x3_4
BIN_AMD_INFER_FOUR=0 run f16_17 --precision f16
BIN_AMD_INFER_FOUR=1 run f16_4 --precision f16
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2p_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:40s} {d['value']:8.3f} {d['ms_per_step']:8.2f} ms")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

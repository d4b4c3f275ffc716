#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 "$@" 2>&1 | tail -1 ) > gpurun_out/r2k_$tag.log 2>&1; }
for rep in 1 2; do
run s1_$rep --streams 1
run s2_$rep --streams 2
run s3_$rep --streams 3
run batched_$rep --streams 2 --batched
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2k_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:45s} {d['value']:8.3f} {d['ms_per_step']:8.2f} ms")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 "$@" 2>&1 | tail -1 ) > gpurun_out/r2f_$tag.log 2>&1; }
for rep in 1 2; do
run new_s3_$rep --streams 3
run oldtail_s3_$rep --streams 3 --variant tail=1
run shape2_s3_$rep --streams 3 --variant 0=2 --variant 4=2
run shape3_s3_$rep --streams 3 --variant 0=3 --variant 4=3
run allold_s3_$rep --streams 3 --variant tail=1 --variant 0=1 --variant 4=1
run new_s1_$rep --streams 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2f_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:45s} {d['value']:8.3f} fps {d['ms_per_step']:8.2f} ms  rdbconv {d['roofline']['avg_kernel_us'] if d['roofline'] else None}")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

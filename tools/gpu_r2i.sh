#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp


export BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 30 "$@" 2>&1 | tail -1 ) > gpurun_out/r2i_$tag.log 2>&1; }
for rep in 1 2; do
run wide_$rep
run narrow_$rep --variant=-4=0
run wide_train_$rep --mode train --steps 6
run narrow_train_$rep --mode train --steps 6 --variant=-4=0
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2i_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:45s} {d['value']:8.3f} {d['unit'][:12]} {d['ms_per_step']:8.2f} ms")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) > gpurun_out/r2l_pytest.log 2>&1
tail -3 gpurun_out/r2l_pytest.log
export BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so
run() { tag=$1; shift; ( timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 ) > gpurun_out/r2l_$tag.log 2>&1; }
for rep in 1 2; do
run train_big_$rep --mode train --steps 8 --variant=7=0
run train_small_$rep --mode train --steps 8 --variant=7=1
run infer_big_$rep --steps 30 --variant=7=0
run infer_small_$rep --steps 30 --variant=7=1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2l_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f"{f:45s} {d['value']:8.3f} {d['ms_per_step']:8.2f} ms")
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

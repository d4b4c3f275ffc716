#!/bin/bash
# round-3 check: new GPU tests first (fail fast), then the whole -m gpu suite, a bench line with the new fields, the
# fp16-headroom table.  usage: gpurun -- bash tools/gpu_r3_check.sh [quick]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=gpurun_out/r3a
( timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -s 2>&1 | tail -40 ) > ${T}_pytest_r3.log 2>&1
tail -25 ${T}_pytest_r3.log
if [ "$1" != "quick" ]; then
  ( timeout 1200 python -m pytest tests -q -m gpu --deselect tests/test_gpu_round3.py 2>&1 | grep -E "passed|failed|rror|^E |^FAILED" | tail -12 ) > ${T}_pytest_all.log 2>&1
  cat ${T}_pytest_all.log
fi
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 ) > ${T}_bench.json 2>${T}_bench.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r3a_bench.json").read())
    r = d["roofline"]; t = d["train"]; tm = d["tolerance_mode"]
    print("fps", d["value"], "ms", d["ms_per_step"], "kern_us", r["avg_kernel_us"], "frac", r["frac"], "power", d["power"])
    print("f16", tm["value"], tm["roofline"]["avg_kernel_us"] if tm.get("roofline") else None, tm["power"])
    print("train", t.get("ms_per_step"), t.get("power"), json.dumps(t.get("roofline"))[:900] if isinstance(t, dict) else t)
    print("stream", d["streaming"])
except Exception as e:
    print("bench parse failed", e); print(open("gpurun_out/r3a_bench.json").read()[-2000:]); print(open("gpurun_out/r3a_bench.err").read()[-2000:])
PY
( timeout 600 python tools/fp16_headroom.py --steps 200 --out gpurun_out/r03_fp16_headroom 2>&1 | tail -60 ) > ${T}_headroom.log 2>&1
grep -E "Worst|Error|error" ${T}_headroom.log | head -20

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py tests/test_gpu_backward.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r2d_pytest.log 2>&1
tail -4 gpurun_out/r2d_pytest.log
( BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so timeout 300 python tools/bench_layers.py --nterms 3 --classes 9 2>&1 | tail -6 ) > gpurun_out/r2d_layers.log 2>&1
cat gpurun_out/r2d_layers.log
for st in 1 2 3 4; do
  ( timeout 300 python bench.py --no-cpu-baseline --no-extras --streams $st 2>&1 | tail -1 ) > gpurun_out/r2d_bench_s$st.log 2>&1
done
( timeout 300 python bench.py --no-cpu-baseline --no-extras --batched 2>&1 | tail -1 ) > gpurun_out/r2d_bench_batched.log 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2d_bench_*.log")):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"] if d["roofline"] else None)
        elif "rror" in ln:
            print(f, ln.strip()[:200])
PY

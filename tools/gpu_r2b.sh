#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so timeout 300 python tools/bench_layers.py --nterms 3 --classes 0,4 2>&1 | tail -20 ) > gpurun_out/r2b_layers_x3.log 2>&1
( BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so timeout 300 python tools/bench_layers.py --nterms 3 --classes 0 --n 4 2>&1 | tail -12 ) > gpurun_out/r2b_layers_x3_n4.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r2b_pytest.log 2>&1
( timeout 300 python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -2 ) > gpurun_out/r2b_bench.log 2>&1
( timeout 300 python bench.py --no-cpu-baseline --no-extras --batched 2>&1 | tail -2 ) > gpurun_out/r2b_bench_batched.log 2>&1
cat gpurun_out/r2b_layers_x3.log gpurun_out/r2b_layers_x3_n4.log; tail -4 gpurun_out/r2b_pytest.log
python - <<'PY'
import json
for f in ("gpurun_out/r2b_bench.log", "gpurun_out/r2b_bench_batched.log"):
    for ln in open(f):
        if ln.startswith("{"):
            d = json.loads(ln)
            print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"] if d["roofline"] else None)
PY

#!/bin/bash
# round 3: the two-rows-per-stage rolling weight-gradient kernel (tuning build, flag 64) against the eight-wave default:
# correctness (backward tests on the side build with the flag), isolated layers, whole training step — one box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T=gpurun_out/r3w
SIDE=tools/_abl/libbinhip_tuning.so
( BIN_AMD_LIB=$SIDE BIN_AMD_WG_DEBUG=64 timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round2.py -q -x 2>&1 | grep -E "passed|failed|rror|^E " | tail -8 ) 2>&1 | tee ${T}_pytest.log
BIN_AMD_LIB=$SIDE WG_DBGS=0,64,65,66,128 WG_LAYERS="3,96,32;3,128,32;3,160,32;3,192,32;3,96,96" timeout 300 python tools/bench_wgrad.py 2>&1 | tail -6 | tee ${T}_layers.log
for rep in 1 2; do
  for v in 0 64; do
    echo "== wg_debug $v rep $rep"
    ( BIN_AMD_LIB=$SIDE BIN_AMD_WG_DEBUG=$v timeout 300 python bench.py --mode train --steps 6 --warmup 2 2>&1 | tail -1 ) > ${T}_train_${v}_${rep}.json
    python - <<PY
import json
d = json.loads(open("${T}_train_${v}_${rep}.json").read())
k = d["roofline"].get("dominant_kernel", {})
print(d["ms_per_step"], d["value"], "wgrad us", k.get("avg_kernel_us"), "beside", k.get("avg_kernel_us_beside_backward_data"), d["power"]["clock_mhz"], d["power"]["power_w"])
PY
  done
done 2>&1 | tee ${T}_train.log

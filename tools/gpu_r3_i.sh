#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round2.py -q -x 2>&1 | grep -E "passed|failed|rror|^E |^FAILED" | tail -5 )
for rep in 1 2 3; do
  for v in wgstride product; do
    if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=tools/_abl/libbinhip_$v.so; fi
    echo -n "== $v $rep: "
    ( timeout 300 python bench.py --mode train --steps 6 --warmup 2 2>&1 | tail -1 ) > /tmp/t.json
    python - <<'PY'
import json
d = json.loads(open("/tmp/t.json").read())
k = d["roofline"]["dominant_kernel"]
print(d["ms_per_step"], "wgrad us", k["avg_kernel_us"], "beside", k["avg_kernel_us_beside_backward_data"])
PY
  done
done 2>&1 | tee gpurun_out/r3i_wgwalk.log
unset BIN_AMD_LIB
PT="python bench.py --mode train --batch 8 --steps 1 --warmup 1"
rm -rf /tmp/pmc_tf /tmp/pmc_tw
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tf -- $PT > /dev/null 2>&1
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tw -- $PT > /dev/null 2>&1
python tools/pmc_traffic.py /tmp/pmc_tf /tmp/pmc_tw | grep -E "wgrad|kernel" | head -6

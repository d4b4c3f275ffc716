#!/bin/bash
# same-box A/B of a side build against the product library: gpu_ab_lib.sh <side .so> [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
SIDE=$1; shift
( BIN_AMD_LIB=$SIDE timeout 600 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py -q -x 2>&1 | grep -E "passed|failed|rror|^E " | tail -5 ) 2>&1 | tee gpurun_out/ab_pytest.log
for rep in 1 2; do
  for v in product side; do
    if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=$SIDE; fi
    echo "== $v $rep"
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 "$@" 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_kernel_us": [0-9.]*' | head -3 | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/ab_bench.log

#!/bin/bash
# same-box A/B: product vs progress-ordered wave priority in the plane-split conv (prio1) and also in the fused tail (prio3)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for side in tools/_abl/libbinhip_prio1.so tools/_abl/libbinhip_prio3.so; do
  ( BIN_AMD_LIB=$side timeout 600 python -m pytest tests/test_gpu_net.py -q -x 2>&1 | grep -E "passed|failed|rror|^E " | tail -3 )
done
for rep in 1 2 3; do
  for v in product prio1 prio3; do
    if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=tools/_abl/libbinhip_$v.so; fi
    echo -n "== $v $rep: "
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_kernel_us": [0-9.]*' | head -3 | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/r3p_fwd.log
for rep in 1 2; do
  for v in product prio3; do
    if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=tools/_abl/libbinhip_$v.so; fi
    echo -n "== train $v $rep: "
    timeout 300 python bench.py --mode train --steps 6 --warmup 2 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
done 2>&1 | tee gpurun_out/r3p_train.log

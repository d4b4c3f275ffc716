#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_backward.py tests/test_gpu_net.py -x -q 2>&1 | grep -E "passed|failed|rror|^E |^FAILED" | tail -8 )
for rep in 1 2; do
  for v in 1 0; do
    echo -n "== train relayout_batch=$v rep $rep: "
    BIN_AMD_RELAYOUT_BATCH=$v timeout 300 python bench.py --mode train --steps 8 --warmup 2 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
done 2>&1 | tee gpurun_out/r3c_train.log

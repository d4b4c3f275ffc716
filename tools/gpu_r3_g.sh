#!/bin/bash
# epilogue rewrite (scalar bias loads, no loads between stores, grouped extras): tests, then same-box A/B against the library built from the previous commit
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|rror|^E |^FAILED" | tail -8 ) 2>&1 | tee gpurun_out/r3g_pytest.log
for rep in 1 2 3; do
  for v in before product; do
    if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=tools/_abl/libbinhip_$v.so; fi
    echo -n "== $v $rep: f16x3 "
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*\|"avg_kernel_us": [0-9.]*' | head -2 | tr '\n' ' '
    echo -n " f16 "
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --precision f16 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '
    echo -n " train "
    timeout 300 python bench.py --mode train --steps 6 --warmup 2 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1
  done
done 2>&1 | tee gpurun_out/r3g_ab.log

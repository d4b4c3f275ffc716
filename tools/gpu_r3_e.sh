#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_backward.py -x -q 2>&1 | grep -E "passed|failed|rror|^E |^FAILED" | tail -8 )
for rep in 1 2 3; do
  for v in product nodot2; do
    if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=tools/_abl/libbinhip_$v.so; fi
    echo -n "== $v $rep: "
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '
    timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --precision f16 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '; echo
  done
done 2>&1 | tee gpurun_out/r3e_dot2.log
unset BIN_AMD_LIB
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r3e -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python tools/stats_md.py gpurun_out/prof_r3e 12 | tee gpurun_out/r3e_stats.md
find gpurun_out/prof_r3e -name "*kernel_trace.csv" -delete

#!/bin/bash
# does filling the empty wave slots of one launch with another call's workgroups (2-3 streams) buy time at the power cap?
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for st in 1 2 3; do
  echo -n "streams $st: "
  timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 --streams $st 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['power']
print(d['ms_per_step'], 'ms', p['clock_mhz']['mean'], 'MHz', p['power_w']['mean'], 'W', 'kernel', d['roofline']['avg_kernel_us'])"
done; done | tee gpurun_out/r3_streams.log

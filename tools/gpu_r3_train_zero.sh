#!/bin/bash
# is the training step clock-limited by the power cap like the forward?  real data vs all-zero operands (same launches)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do for z in "" "--zero-data"; do
  echo -n "train ${z:-real}: "
  timeout 300 python bench.py --mode train --steps 10 $z 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['power']; k=d['roofline']['dominant_kernel']
print(d['ms_per_step'], 'ms', p['clock_mhz']['mean'], 'MHz', p['power_w']['mean'], 'W', 'wgrad3x3', k['avg_kernel_us'], 'us, beside', k['avg_kernel_us_beside_backward_data'])"
done; done | tee gpurun_out/r3_train_zero.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_r3_wgrad.sh
echo "==== s_setprio side build (forward 720p)"
bash tools/gpu_ab_lib.sh tools/_abl/libbinhip_prio.so
echo "==== small ATen ops of one training step"
( timeout 300 python tools/profile_train_ops.py 2>&1 | tail -80 ) > gpurun_out/r3b_train_ops.log; head -60 gpurun_out/r3b_train_ops.log

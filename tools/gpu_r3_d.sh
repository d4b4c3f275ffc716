#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -s -k "constructor or unsupported or relayout" 2>&1 | grep -vE "^\s*$|Conv2d|Sequential|ReLU|\)$" | tail -40 ) 2>&1 | tee gpurun_out/r3d_shapes.log
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|rror|^E |^FAILED" | tail -12 ) 2>&1 | tee gpurun_out/r3d_all.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_round2.py -m gpu -q 2>&1 | tail -60 ) > gpurun_out/r2h_pytest_new.log 2>&1
tail -60 gpurun_out/r2h_pytest_new.log

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1000 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|rror|^E " | tail -4
( timeout 300 python bench.py --mode train 2>&1 | tail -1 ) | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | tr '\n' ' '; echo
( timeout 300 python bench.py --precision f16 --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -1 ) | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '; echo
( timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -1 ) | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '; echo
bash tools/gpu_pmc_train.sh | grep "conv_mfma_kernel<1, 6\|conv_mfma_kernel<5\|conv_mfma_kernel<1, 7\|conv_mfma_kernel<1, 3"

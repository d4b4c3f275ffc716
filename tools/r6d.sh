#!/bin/bash
# round 6, GPU call D: packed hi/lo split in the epilogues — bit test, full suite, A/B against the scalar-split side build (window + training step)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== split test"; python -m pytest tests/test_gpu_conv.py -m gpu -q -x -k "packed_split or saturation or reserved" 2>&1 | tail -5
echo "=== full suite"; python -m pytest tests -m gpu -q -x 2>&1 | tail -4
echo "=== A/B window (product = packed split, side = scalar split)"; bash tools/gpu.sh "tag r6d" "ab tools/_abl/libbinhip_scalar_split.so" 2>&1 | tail -6
echo "=== A/B train"; bash tools/gpu.sh "tag r6dt" "ab tools/_abl/libbinhip_scalar_split.so --mode train --batch 8 --no-power" 2>&1 | tail -6

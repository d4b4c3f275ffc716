#!/bin/bash
# round 6, GPU call C: DMA-spread A/B, zero-operand timeline with the atomic-free recorder, headroom with trained-like weights, new tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "=== new tests"; python -m pytest tests/test_gpu_net.py -m gpu -q -k "small_input or trained_like" 2>&1 | tail -3
echo "=== A/B spread DMA"; bash tools/gpu.sh "tag r6c" "ab tools/_abl/libbinhip_spread.so" 2>&1 | tail -8
echo "=== golden with spread"; BIN_AMD_LIB=tools/_abl/libbinhip_spread.so python -m pytest tests/test_gpu_net.py -m gpu -q -k "golden" 2>&1 | tail -2
echo "=== timeline zero"; BIN_AMD_LIB=tools/_abl/libbinhip_timeline.so timeout 600 python tools/wg_timeline.py --zero --out gpurun_out/r6c_tl_zero > gpurun_out/r6c_tl_zero.log 2>&1; tail -3 gpurun_out/r6c_tl_zero.log
echo "=== headroom trained-like"; timeout 900 python tools/fp16_headroom.py --trained-like 0 --steps 0 --out gpurun_out/r6c_headroom_trained_like > gpurun_out/r6c_headroom.log 2>&1; tail -12 gpurun_out/r6c_headroom.log

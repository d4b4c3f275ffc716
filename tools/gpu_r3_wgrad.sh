#!/bin/bash
# 3x3 weight gradient: backward tests on the product library, then the tuning builds taken apart —
# 0 product (wave = X row, DMA spread over the multiply steps), 0x10000 the round-2 kernel (wave = gY row, burst);
# 1 no DMA, 8 no fragment reads, 9 MFMAs alone, 2 DMA alone; libbinhip_tuning_burst.so = wave = X row with the DMA burst
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_round2.py -q -x 2>&1 | grep -E "passed|failed|rror|^E " | tail -6 ) 2>&1 | tee gpurun_out/r3_wgrad_pytest.log
for rep in 1 2; do
for lib in tuning tuning_burst; do
echo "== $lib"
BIN_AMD_LIB=tools/_abl/libbinhip_$lib.so WG_DBGS=${WG_DBGS:-0,0x10000,8} WG_LAYERS="3,96,32;3,128,32;3,160,32;3,192,32" timeout 300 python tools/bench_wgrad.py 2>&1 | grep -v amdgpu.ids
done
done | tee gpurun_out/r3_wgrad_parts.log

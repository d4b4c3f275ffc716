#!/bin/bash
# round 6, GPU call F: MFMA power probe with coarser lo-plane mantissas; patch-DMA nt A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for m in 0 4 17 12 13 15 16 3 2; do timeout 60 ./tools/probe_mfma_power $m 4 ; done 2>&1 | tee gpurun_out/r6f_probe.log
echo "=== A/B patch nt"; bash tools/gpu.sh "tag r6f" "ab tools/_abl/libbinhip_patch_nt.so" 2>&1 | tail -6

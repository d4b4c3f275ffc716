#!/bin/bash
# per-kernel HBM-side traffic of the training step (two PMC passes, as for the forward)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
P="python bench.py --mode train --batch 8 --steps 1 --warmup 1"
rm -rf /tmp/pmc_tf /tmp/pmc_tw
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tf -- $P > /dev/null 2>&1
BIN_AMD_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_tw -- $P > /dev/null 2>&1
python tools/pmc_traffic.py /tmp/pmc_tf /tmp/pmc_tw > gpurun_out/pmc_train.md 2>&1
head -24 gpurun_out/pmc_train.md

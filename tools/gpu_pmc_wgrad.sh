#!/bin/bash
# SQ counters of one weight-gradient layer (160 -> 32, 3x3, 40 x 128 x 128): gpu_pmc_wgrad.sh [kernel-name-substring]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
K=${1:-wgrad3x3}
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
rm -rf /tmp/pmc_a /tmp/pmc_b
export WG_LAYERS="3,160,32" WG_DBGS=${WG_DBGS:-0}
timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/bench_wgrad.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/bench_wgrad.py > /dev/null 2>&1
( python tools/pmc_sum.py /tmp/pmc_a $K ; python tools/pmc_sum.py /tmp/pmc_b $K ) 2>&1 | tee gpurun_out/pmc_wgrad_$K.log

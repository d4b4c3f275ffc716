#!/bin/bash
# round 3: SQ / GRBM counters of the dominant fp32-class kernels (dense-block conv, fused tail, 3x3 weight gradient), two
# 8-counter passes each, no other trace domains.   usage: gpurun -- bash tools/gpu_pmc_sq_r3.sh
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
for what in "rdb 3 160" "tail 3 192" "wgrad 3 160"; do
  rm -rf /tmp/pmc_a /tmp/pmc_b
  timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/pmc_one.py $what > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/pmc_one.py $what > /dev/null 2>&1
  echo "=== pmc_one.py $what"
  case "$what" in wgrad*) F=wgrad3x3;; rdb*) F=conv_x3;; *) F=_x3_;; esac
  python tools/pmc_sum.py /tmp/pmc_a $F ; python tools/pmc_sum.py /tmp/pmc_b $F
done > gpurun_out/r3_pmc_sq.log 2>&1
cat gpurun_out/r3_pmc_sq.log

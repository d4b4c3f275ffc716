#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
export BIN_AMD_LIB=tools/_abl/libbinhip_tuning.so
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
for v in 2 3 1; do
  rm -rf /tmp/pmc_a /tmp/pmc_b
  timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/pmc_one.py rdb 3 192 $v > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/pmc_one.py rdb 3 192 $v > /dev/null 2>&1
  echo "=== variant $v (RDB conv 192->32, f16x3)"
  python tools/pmc_sum.py /tmp/pmc_a conv_ ; python tools/pmc_sum.py /tmp/pmc_b conv_
done > gpurun_out/r2c_pmc_x3.log 2>&1
cat gpurun_out/r2c_pmc_x3.log

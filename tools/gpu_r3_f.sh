#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 python tools/probe_hbm_rw.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3f_hbm_rw.log
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
for lib in product prio1; do
  if [ $lib = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=tools/_abl/libbinhip_$lib.so; fi
  rm -rf /tmp/pmc_a /tmp/pmc_b
  timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/pmc_one.py rdb 3 160 > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/pmc_one.py rdb 3 160 > /dev/null 2>&1
  echo "=== $lib: pmc_one.py rdb 3 160"
  python tools/pmc_sum.py /tmp/pmc_a _x3_ ; python tools/pmc_sum.py /tmp/pmc_b _x3_
done 2>&1 | tee gpurun_out/r3f_pmc_sq.log

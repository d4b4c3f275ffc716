#!/bin/bash
# round-2 profile collection: kernel-trace stats of the default bench command, per-precision PMC traffic passes,
# SQ counters of the two dominant fp32-class kernels, power / clock log of the timed region
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
rm -rf gpurun_out/prof_r2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_x3 -o kt -- $B > gpurun_out/prof_r2_kt_x3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_f16 -o kt -- $B --precision f16 > gpurun_out/prof_r2_kt_f16.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r2/kt_train -o kt -- python bench.py --mode train --batch 8 --steps 2 --warmup 1 > gpurun_out/prof_r2_kt_train.log 2>&1
python tools/stats_md.py gpurun_out/prof_r2/kt_x3 16 > gpurun_out/r2j_stats_x3.md
python tools/stats_md.py gpurun_out/prof_r2/kt_f16 12 > gpurun_out/r2j_stats_f16.md
python tools/stats_md.py gpurun_out/prof_r2/kt_train 26 > gpurun_out/r2j_stats_train.md
P="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --calib"
rm -f gpurun_out/r2j_pmc_traffic.json
for prec in f16x3 f16; do
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f_$prec -- $P --precision $prec > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w_$prec -- $P --precision $prec > /dev/null 2>&1
  python tools/pmc_traffic.py /tmp/pmc_f_$prec /tmp/pmc_w_$prec --json gpurun_out/r2j_pmc_traffic.json --key $prec > gpurun_out/r2j_pmc_traffic_$prec.md
done
find gpurun_out/prof_r2 -name "*kernel_trace.csv" -delete
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
for what in "rdb 3 160" "tail 3 192"; do
  rm -rf /tmp/pmc_a /tmp/pmc_b
  timeout 200 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/pmc_one.py $what > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/pmc_one.py $what > /dev/null 2>&1
  echo "=== pmc_one.py $what"
  python tools/pmc_sum.py /tmp/pmc_a _x3_ ; python tools/pmc_sum.py /tmp/pmc_b _x3_
done > gpurun_out/r2j_pmc_sq.log 2>&1
tools/smi_watch.sh gpurun_out/r2j_smi_x3.log -- timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 > gpurun_out/r2j_bench_x3.log 2>&1
tools/smi_watch.sh gpurun_out/r2j_smi_x3_zero.log -- timeout 300 python bench.py --no-cpu-baseline --no-extras --steps 60 --zero-data > gpurun_out/r2j_bench_x3_zero.log 2>&1
tools/smi_watch.sh gpurun_out/r2j_smi_train.log -- timeout 300 python bench.py --mode train --steps 12 > gpurun_out/r2j_bench_train.log 2>&1
cat gpurun_out/r2j_stats_x3.md | head -12; cat gpurun_out/r2j_stats_train.md; head -3 gpurun_out/r2j_pmc_traffic_f16x3.md; cat gpurun_out/r2j_pmc_sq.log
for t in x3 x3_zero train; do echo "== $t"; grep -o '"value": [0-9.]*, "unit": "[a-z /]*", "n_gpus": 1, "steps": [0-9]*[^}]*"ms_per_step": [0-9.]*' gpurun_out/r2j_bench_$t.log | head -1; grep -c . gpurun_out/r2j_smi_$t.log; sort -t'(' -k3 gpurun_out/r2j_smi_$t.log | tail -4; done

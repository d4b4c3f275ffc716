#!/bin/bash
# matrix-core throughput at the power cap per MFMA shape / operand data (tools/probe_mfma_power.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for m in ${MODES:-0 1 2 3 4 5 6 0 1}; do
  tools/smi_watch.sh gpurun_out/mfma_smi_$m.log -- timeout 60 tools/probe_mfma_power $m 4 > gpurun_out/mfma_probe_$m.log 2>&1
  cat gpurun_out/mfma_probe_$m.log
  sed 's/GPU\[0\]\t\t: //g' gpurun_out/mfma_smi_$m.log | awk '{gsub(/[()Mhz]/,"",$5); if ($NF+0 > 600) {c+=$5; p+=$NF; n++}} END {if (n) printf "   busy samples %d: %.0f MHz, %.0f W\n", n, c/n, p/n}'
  sleep 2
done

#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for v in product abl1 abl2 abl4 zero; do
  if [ $v = product ]; then unset BIN_AMD_LIB; A=""; elif [ $v = zero ]; then unset BIN_AMD_LIB; A="zero"; else export BIN_AMD_LIB=tools/_abl/libbinhip_$v.so; A=""; fi
  tools/smi_watch.sh gpurun_out/r2q_smi_$v.log -- timeout 120 python tools/power_probe.py 4 $A > gpurun_out/r2q_probe_$v.log 2>&1
  echo "== $v"; grep launches gpurun_out/r2q_probe_$v.log
  sed 's/GPU\[0\]\t\t: //g' gpurun_out/r2q_smi_$v.log | awk '{gsub(/[()Mhz]/,"",$5); if ($NF+0 > 600) {c+=$5; p+=$NF; n++}} END {if (n) printf "busy samples %d: %.0f MHz, %.0f W\n", n, c/n, p/n}'
done

#!/bin/bash
# round 6, GPU call B: static-ownership dense block (BIN_AMD_RDB3=1) — bit check, A/B, timeline; stall reproduction
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TL=tools/_abl/libbinhip_timeline.so
echo "=== check_rdb3"; timeout 900 python tools/check_rdb3.py > gpurun_out/r6b_check.log 2>&1; tail -9 gpurun_out/r6b_check.log
echo "=== A/B window"
for rep in 1 2 3; do for v in 0 1; do
  echo -n "RDB3=$v rep $rep: "
  BIN_AMD_RDB3=$v timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 20 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"cycles_per_step_M": [0-9.]*\|"xcd_clock_mhz_mean": [0-9.]*' | head -4 | tr '\n' ' '; echo
done; done 2>&1 | tee gpurun_out/r6b_ab.log
echo "=== timeline product"; BIN_AMD_LIB=$TL timeout 600 python tools/wg_timeline.py --out gpurun_out/r6b_tl_real > gpurun_out/r6b_tl_real.log 2>&1; tail -12 gpurun_out/r6b_tl_real.log
echo "=== timeline rdbs"; BIN_AMD_LIB=$TL timeout 600 python tools/wg_timeline.py --plan-flags 4 --out gpurun_out/r6b_tl_rdbs > gpurun_out/r6b_tl_rdbs.log 2>&1; tail -12 gpurun_out/r6b_tl_rdbs.log
echo "=== stall reproduction (side stream off, as the r05 trace)"
for k in 1 2; do
  rm -rf /tmp/stall$k; cd /tmp
  BIN_AMD_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/stall$k -o st -- python $GRAFT_REPO_ROOT/bench.py --mode train --batch 8 --steps 2 --warmup 1 --no-power > $GRAFT_REPO_ROOT/gpurun_out/r6b_trace$k.log 2>&1
  cd $GRAFT_REPO_ROOT
  python tools/stall_hunt.py trace /tmp/stall$k > gpurun_out/r6b_stalls$k.json 2> gpurun_out/r6b_stalls$k.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r6b_stalls$k.json"))
for s in d["stalls"][:4]:
    print(s["kernel"][:60], "#%d/%d"%(s["launch_of_this_kernel"],s["of"]), s["us"], "med", s["median_us"], "t", s["ms_after_first_dispatch"], [(a["fn"],a["overlap_us"],a["call_us"]) for a in s["host_api_overlap"][:3]])
PY
done

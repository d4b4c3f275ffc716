#!/bin/bash
# round 6, GPU call A: workgroup timeline (real / zero / rdb3), per-step jitter, stall hunt under rocprofv3
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TL=tools/_abl/libbinhip_timeline.so
echo "=== timeline real"; BIN_AMD_LIB=$TL timeout 600 python tools/wg_timeline.py --out gpurun_out/r6a_tl_real > gpurun_out/r6a_tl_real.log 2>&1; tail -5 gpurun_out/r6a_tl_real.log
echo "=== timeline zero"; BIN_AMD_LIB=$TL timeout 600 python tools/wg_timeline.py --zero --out gpurun_out/r6a_tl_zero > gpurun_out/r6a_tl_zero.log 2>&1; tail -3 gpurun_out/r6a_tl_zero.log
echo "=== timeline rdb3"; BIN_AMD_LIB=$TL timeout 600 python tools/wg_timeline.py --plan-flags 4 --out gpurun_out/r6a_tl_rdb3 > gpurun_out/r6a_tl_rdb3.log 2>&1; tail -3 gpurun_out/r6a_tl_rdb3.log
echo "=== steps train"; timeout 900 python tools/stall_hunt.py steps --steps 60 --mode train > gpurun_out/r6a_steps_train.json 2> gpurun_out/r6a_steps_train.err; cut -c1-600 gpurun_out/r6a_steps_train.json; tail -2 gpurun_out/r6a_steps_train.err
echo "=== steps infer"; timeout 900 python tools/stall_hunt.py steps --steps 40 --mode infer > gpurun_out/r6a_steps_infer.json 2> gpurun_out/r6a_steps_infer.err; cut -c1-600 gpurun_out/r6a_steps_infer.json; tail -2 gpurun_out/r6a_steps_infer.err
echo "=== trace"
rm -rf /tmp/stall; cd /tmp
timeout 900 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/stall -o st -- python $GRAFT_REPO_ROOT/bench.py --mode train --batch 8 --steps 2 --warmup 1 --no-power > $GRAFT_REPO_ROOT/gpurun_out/r6a_trace.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/r6a_trace.log | cut -c1-300
python tools/stall_hunt.py trace /tmp/stall > gpurun_out/r6a_stalls.json 2> gpurun_out/r6a_stalls.err; head -c 3000 gpurun_out/r6a_stalls.json; tail -3 gpurun_out/r6a_stalls.err

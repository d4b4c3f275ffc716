#!/bin/bash
# One parameterised runner for everything this repo does on the GPU box (replaces the per-session gpu_*.sh scripts of
# rounds 1-3).  Every argument is one STEP ("<verb> [args]"); steps run in order, logs go to gpurun_out/<tag>_*.
#
#   gpurun --timeout 1500 -- bash tools/gpu.sh "tag r4a" "test tests/test_gpu_net.py -x" "ab tools/_abl/libbinhip_r3.so"
#
# verbs
#   tag NAME                    prefix of the log files that follow (default: run)
#   test [pytest args]          python -m pytest -m gpu -q [args]         (default: the whole tests/ directory)
#   smoke                       __graft_entry__.smoke()
#   bench [bench.py args]       one bench line -> <tag>_bench[_N].json, key figures printed
#   ab SIDE.so [bench.py args]  same-box A/B, three alternating repetitions: product library vs BIN_AMD_LIB=SIDE.so
#   kt NAME [bench.py args]     rocprofv3 --kernel-trace --stats of the bench command -> gpurun_out/prof/<NAME> + <tag>_stats_<NAME>.md
#   traffic KEY [bench.py args] two PMC passes (FETCH_SIZE, WRITE_SIZE; nothing but --kernel-trace beside them) ->
#                               <tag>_pmc_traffic.json[KEY] + <tag>_pmc_traffic_<KEY>.md
#   sq "rdb 3 160"              SQ counters (two 8-counter passes) of one layer class of tools/pmc_one.py
#   py SCRIPT [args]            python SCRIPT args, output to <tag>_<script>.log (tail printed)
#   tl NAME [wg_timeline args]  per-workgroup timeline (tools/wg_timeline.py) with the BINHIP_TIMELINE side build -> <tag>_tl_<NAME>.json/.npz
#   steps MODE N                un-profiled per-step spread (tools/stall_hunt.py steps): MODE = train | infer -> <tag>_steps_<MODE>.json
#   stalls NAME [bench args]    rocprofv3 --kernel-trace --hip-trace of the bench command + tools/stall_hunt.py trace -> <tag>_stalls_<NAME>.json
#   env NAME=VALUE              export for the steps that follow (env NAME= unsets)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=run; NB=0; NP=0
keyfig() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{"metric"')][-1])
except Exception as e:
    print("no bench line:", e); print(open(sys.argv[1]).read()[-1500:]); raise SystemExit(0)
r = d.get("roofline", {})
out = {"value": d.get("value"), "ms": d.get("ms_per_step"), "kern_us": r.get("avg_kernel_us"), "frac": r.get("frac"), "bound": r.get("bound")}
for k in ("power", "power_bound", "harness", "streaming"):
    if k in d: out[k] = d[k]
tm, tr = d.get("tolerance_mode"), d.get("train")
if isinstance(tm, dict): out["f16"] = tm.get("value")
if isinstance(tr, dict): out["train_ms"] = tr.get("ms_per_step"); out["train_power_bound"] = tr.get("power_bound")
if d.get("config", {}).get("workload", "").startswith("train") or "dominant_kernel" in r:
    out["dominant"] = r.get("dominant_kernel")
print(json.dumps(out))
PY
}
for step in "$@"; do
  set -- $step; verb=$1; shift
  echo "=== [$TAG] $verb $*"
  case $verb in
    tag) TAG=$1 ;;
    env) case "$1" in *=) unset "${1%=}" ;; *) export "$1" ;; esac ;;
    tl)
      name=$1; shift
      BIN_AMD_LIB=tools/_abl/libbinhip_timeline.so timeout 600 python tools/wg_timeline.py --out gpurun_out/${TAG}_tl_$name "$@" > gpurun_out/${TAG}_tl_$name.log 2>&1
      tail -4 gpurun_out/${TAG}_tl_$name.log ;;
    steps)
      timeout 900 python tools/stall_hunt.py steps --mode $1 --steps ${2:-50} > gpurun_out/${TAG}_steps_$1.json 2> gpurun_out/${TAG}_steps_$1.err
      cut -c1-300 gpurun_out/${TAG}_steps_$1.json ;;
    stalls)
      name=$1; shift; rm -rf /tmp/stall_$name
      ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/stall_$name -o st -- python $OLDPWD/bench.py "$@" > $OLDPWD/gpurun_out/${TAG}_stalls_$name.log 2>&1 )
      python tools/stall_hunt.py trace /tmp/stall_$name > gpurun_out/${TAG}_stalls_$name.json 2> gpurun_out/${TAG}_stalls_$name.err
      head -c 600 gpurun_out/${TAG}_stalls_$name.json ;;
    test)
      args="$*"; [ -z "$args" ] && args=tests
      ( timeout 2400 python -m pytest -m gpu -q $args 2>&1 | grep -E "passed|failed|rror|^E |^FAILED|^tests/.*(FAIL|ERR)" | tail -25 ) 2>&1 | tee -a gpurun_out/${TAG}_pytest.log ;;
    smoke)
      ( timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4 ) | tee gpurun_out/${TAG}_smoke.log ;;
    bench)
      NB=$((NB+1)); f=gpurun_out/${TAG}_bench_$NB.json
      ( time timeout 1200 python bench.py "$@" 2>gpurun_out/${TAG}_bench_$NB.err | tail -1 ) > $f 2> gpurun_out/${TAG}_bench_$NB.time
      grep real gpurun_out/${TAG}_bench_$NB.time; keyfig $f; tail -3 gpurun_out/${TAG}_bench_$NB.err ;;
    ab)
      side=$1; shift
      for rep in 1 2 3; do for v in product side; do
        if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=$side; fi
        echo -n "$v $rep: "
        timeout 400 python bench.py --no-cpu-baseline --no-extras --steps 20 "$@" 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"avg_kernel_us": [0-9.]*' | head -3 | tr '\n' ' '; echo
      done; done 2>&1 | tee -a gpurun_out/${TAG}_ab.log
      unset BIN_AMD_LIB ;;
    kt)
      name=$1; shift; rm -rf gpurun_out/prof/$name
      timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/$name -o kt -- python bench.py "$@" > gpurun_out/${TAG}_kt_$name.log 2>&1
      python tools/stats_md.py gpurun_out/prof/$name 24 > gpurun_out/${TAG}_stats_$name.md
      find gpurun_out/prof/$name -name "*kernel_trace.csv" -delete
      head -16 gpurun_out/${TAG}_stats_$name.md ;;
    traffic)
      key=$1; shift; rm -rf /tmp/pmc_f /tmp/pmc_w
      timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -- python bench.py "$@" > /dev/null 2>&1
      timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -- python bench.py "$@" > /dev/null 2>&1
      python tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w --json gpurun_out/${TAG}_pmc_traffic.json --key $key > gpurun_out/${TAG}_pmc_traffic_$key.md
      head -12 gpurun_out/${TAG}_pmc_traffic_$key.md ;;
    sq)
      P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
      P2="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_SALU GRBM_GUI_ACTIVE"
      rm -rf /tmp/pmc_a /tmp/pmc_b
      timeout 300 rocprofv3 --pmc $P1 --kernel-trace --output-format csv -d /tmp/pmc_a -- python tools/pmc_one.py "$@" > /dev/null 2>&1
      timeout 300 rocprofv3 --pmc $P2 --kernel-trace --output-format csv -d /tmp/pmc_b -- python tools/pmc_one.py "$@" > /dev/null 2>&1
      case "$1" in wgrad*) F=wgrad3x3;; rdb*) F=conv_x3;; *) F=_x3_;; esac
      ( echo "=== pmc_one.py $*"; python tools/pmc_sum.py /tmp/pmc_a $F; python tools/pmc_sum.py /tmp/pmc_b $F ) 2>&1 | tee -a gpurun_out/${TAG}_pmc_sq.log ;;
    py)
      script=$1; shift; b=$(basename $script .py); NP=$((NP+1))
      ( timeout 1200 python $script "$@" 2>&1 | tail -150 ) > gpurun_out/${TAG}_${b}_$NP.log 2>&1; tail -30 gpurun_out/${TAG}_${b}_$NP.log ;;
    *) echo "unknown step: $verb" ;;
  esac
done

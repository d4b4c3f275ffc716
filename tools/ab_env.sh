#!/bin/bash
# Same-box A/B of an ENVIRONMENT switch (three alternating repetitions): tools/ab_env.sh NAME=VALUE [bench.py args]
# e.g.  tools/ab_env.sh BIN_AMD_FUSED_LOSS=0 --mode train --steps 6 --warmup 2 --no-power
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.." || exit 1
kv=$1; shift
for rep in 1 2 3; do for v in default "$kv"; do
  echo -n "$v $rep: "
  if [ "$v" = default ]; then
    timeout 400 python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '
  else
    env "$kv" timeout 400 python bench.py --no-cpu-baseline --no-extras "$@" 2>&1 | tail -1 | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '
  fi
  echo
done; done

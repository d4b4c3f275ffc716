#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
  for v in product prev; do
    if [ $v = product ]; then unset BIN_AMD_LIB; else export BIN_AMD_LIB=tools/_abl/libbinhip_prev.so; fi
    echo "== $v $rep"
    ( timeout 300 python bench.py --precision f16 --no-cpu-baseline --no-extras --steps 30 2>&1 | tail -1 ) | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '
    ( timeout 300 python bench.py --mode train --steps 6 2>&1 | tail -1 ) | grep -o '"ms_per_step": [0-9.]*' | head -1 | tr '\n' ' '; echo
  done
done

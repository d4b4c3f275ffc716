#!/bin/bash
# sample sclk / power while a command runs: smi_watch.sh <outfile> -- cmd...
out=$1; shift; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Average Graphics Package Power|Current Socket Graphics Package Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $out &
W=$!
"$@"
kill $W

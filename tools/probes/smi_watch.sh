#!/bin/bash
# Sample clocks / power while a bench runs: bash tools/smi_watch.sh <out> -- <command ...>
out=$1; shift; shift
( for i in $(seq 1 200); do echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction|hotspot)" ; sleep 0.5; done ) > "$out" 2>&1 &
W=$!
"$@"
rc=$?
kill $W 2>/dev/null
exit $rc

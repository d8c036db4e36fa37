#!/bin/bash
# A/B of a compile-time switch on the GPU box: tools/ab_flags.sh <source.hip> "<flags A>" "<flags B>" <command...>
set -e
src=$1; fa=$2; fb=$3; shift 3
for f in "$fa" "$fb" "$fa" "$fb"; do
  touch dcnet_amd/csrc/$src
  DCN_EXTRA_FLAGS="$f" python -m dcnet_amd.build > /dev/null 2>&1
  echo "== flags: $f"
  "$@"
done
touch dcnet_amd/csrc/$src
python -m dcnet_amd.build > /dev/null 2>&1

#!/bin/bash
# A/B of a compile-time switch on the GPU box: tools/ab_flags.sh <source.hip> "<flags A>" "<flags B>" <command...>
set -e
# whatever happens below, the default build is what is left installed (the build is keyed on its flags: build/FLAGS.stamp)
trap 'env -u DCN_EXTRA_FLAGS python -m dcnet_amd.build > /dev/null 2>&1' EXIT
src=$1; fa=$2; fb=$3; shift 3
for f in "$fa" "$fb" "$fa" "$fb"; do
  DCN_EXTRA_FLAGS="$f" python -m dcnet_amd.build > /dev/null 2>&1
  echo "== flags: $f"
  "$@"
done

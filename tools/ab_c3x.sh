#!/bin/bash
# timing ablations of conv3x.hip (results wrong by construction): one build per bit set, the strip layers' forward / data gradient each
set -e
# whatever happens below, the default build is what is left installed (the build is keyed on its flags: build/FLAGS.stamp)
trap 'env -u DCN_EXTRA_FLAGS python -m dcnet_amd.build > /dev/null 2>&1' EXIT
for abl in ${ABLS:-0 7 5 2 8 15}; do
  DCN_EXTRA_FLAGS="-DC3X_ABL=$abl" python -m dcnet_amd.build > /dev/null 2>&1
  echo "== C3X_ABL=$abl"
  python tools/bench_convs.py --strip --iters 10 --set 3m16=2 2>/dev/null | grep -E "^ +[0-9]+ +[0-9]+ 3 1"
done

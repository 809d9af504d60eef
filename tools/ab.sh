#!/bin/bash
# Same-box A/B of two libshk builds (boxes differ by several %): alternates base/new bench runs.
# usage: tools/ab.sh exp/libshk_prev.so [steps]
BASE=$1; STEPS=${2:-30}
for i in 1 2; do
  SHK_LIB_PATH=$BASE python bench.py --steps $STEPS --warmup 3 --no-cpu 2>/dev/null | python tools/bsum.py base || exit 1
  python bench.py --steps $STEPS --warmup 3 --no-cpu 2>/dev/null | python tools/bsum.py new || exit 1
done

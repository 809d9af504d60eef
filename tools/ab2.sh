#!/bin/bash
# Same-box A/B with extra bench arguments: tools/ab2.sh exp/libshk_prev.so "<bench args>" [rounds]
BASE=$1; ARGS=$2; R=${3:-2}
for i in $(seq $R); do
  SHK_LIB_PATH=$BASE python bench.py --steps 30 --warmup 3 --no-cpu $ARGS 2>/dev/null | python tools/bsum.py base || exit 1
  python bench.py --steps 30 --warmup 3 --no-cpu $ARGS 2>/dev/null | python tools/bsum.py new || exit 1
done

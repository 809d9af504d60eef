#!/bin/bash
# A/B several builds of libshk.so on the default bench: bash tools/ab_libs.sh <tag> <lib or "default"> ...  (two rounds, interleaved)
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
for rnd in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = default ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/exp/libshk_$lib.so; fi
    python3 bench.py --no-extras --no-cpu-baseline --steps 80 --warmup 20 > gpurun_out/${TAG}_${lib}_$rnd.json 2> gpurun_out/${TAG}_${lib}_$rnd.err
    python3 -c "
import json
d=json.load(open('gpurun_out/${TAG}_${lib}_$rnd.json')); print('$lib', $rnd, d['value'], d['kernels_ms_per_step'])"
  done
done

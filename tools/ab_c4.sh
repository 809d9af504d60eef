#!/bin/bash
# A/B builds of libshk.so on the config-4 share (owner exchange over a world of one): bash tools/ab_c4.sh <tag> <reads> <lib or "default"> ...
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
READS=$1; shift
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/exp/libshk_$lib.so; fi
  python3 bench.py --config 4 --reads $READS --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_${lib}.json 2> gpurun_out/${TAG}_${lib}.err || { tail -5 gpurun_out/${TAG}_${lib}.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_${lib}.json').read().strip().splitlines()[-1]); print('$lib', d['value'], d['kernels_ms_per_step'])"
done

#!/bin/bash
# A/B builds of libshk.so on bench.py's "shapes" extras (10 lanes, 30 Mb genome): bash tools/ab_shapes.sh <tag> <lib or "default"> ...
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
for lib in "$@"; do
  if [ "$lib" = default ]; then unset SHK_LIB_PATH; else export SHK_LIB_PATH=$PWD/exp/libshk_$lib.so; fi
  SHK_BENCH_EXTRAS=shapes python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_${lib}.json 2> gpurun_out/${TAG}_${lib}.err || { tail -5 gpurun_out/${TAG}_${lib}.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('gpurun_out/${TAG}_${lib}.json').read().strip().splitlines()[-1])
for k,v in d['extras']['shapes'].items():
    if isinstance(v, dict): print('$lib', k, v.get('Gbases_per_s'), v.get('kernels_ms'))"
done

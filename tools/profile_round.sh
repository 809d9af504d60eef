#!/bin/bash
# Collect the per-round rocprofv3 evidence for bench.py's default (N=1) run on the GPU box:
#   1. --kernel-trace --stats  (per-kernel average durations)
#   2. --pmc FETCH_SIZE        (own pass)
#   3. --pmc WRITE_SIZE        (own pass)
# Outputs land under gpurun_out/<tag>/ ; copy the summaries into profiles/ afterwards.
set -e
TAG=${1:-r01}
ARGS=${2:-"--no-cpu-baseline --no-extras"}  # bench.py's own defaults: the averages in profiles/ are of the same command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py $ARGS > gpurun_out/$TAG/trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_fetch -- python3 bench.py $ARGS > gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_write -- python3 bench.py $ARGS > gpurun_out/$TAG/pmc_write.log 2>&1
echo done

#!/bin/bash
# rocprofv3 PMC passes (two counters per pass, --kernel-trace only) over an arbitrary command; summary per kernel.
#   bash tools/pmc_generic.sh <outdir-under-gpurun_out> <command...>
set -e
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$OUT
i=0
for C in "GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/$OUT/p$i -- "$@" > gpurun_out/$OUT/p$i.log 2>&1 || echo "pass $i failed: $C"
done
python3 tools/pmcsum.py gpurun_out/$OUT > gpurun_out/$OUT/summary.txt

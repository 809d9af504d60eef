#!/bin/bash
# LDS / issue counters of the two hot kernels (experiment helper; separate --pmc passes).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS=${PMC_ARGS:-"--steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-extras"}
mkdir -p gpurun_out/lds
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/lds/p$i -- python3 bench.py $ARGS > gpurun_out/lds/p$i.log 2>&1 || echo "pass $i failed: $C"
done
python3 tools/pmcsum.py gpurun_out/lds > gpurun_out/lds/summary.txt
cat gpurun_out/lds/summary.txt

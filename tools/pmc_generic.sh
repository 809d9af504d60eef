#!/bin/bash
# rocprofv3 PMC passes (two counters per pass, --kernel-trace only) over an arbitrary command; summary per kernel.
#   bash tools/pmc_generic.sh <outdir-under-gpurun_out> <command...>
# The command must START with the interpreter or binary itself (python3 tools/x.py …, ./tools/abench …): the
# profiler's preloaded library initialises the GPU before the program starts, so an `env`, `bash -c`, `timeout`
# or a `#!/usr/bin/env` script in front of it would be an exec from a process that already holds the GPU — which this
# pool refuses (and which can take the machine down).
set -e
OUT=$1; shift
case "$(basename "$1")" in
  python3|python|abench|wbench|ibench|ibench2|shk_count) ;;
  *) echo "pmc_generic.sh: the command must start with python3 or a real binary, not '$1'" >&2; exit 2 ;;
esac
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$OUT
i=0
for C in "GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/$OUT/p$i -- "$@" > gpurun_out/$OUT/p$i.log 2>&1 || echo "pass $i failed: $C"
done
python3 tools/pmcsum.py gpurun_out/$OUT > gpurun_out/$OUT/summary.txt

#!/usr/bin/env python3
"""gpurun_out/lds (written by tools/pmc_lds.sh on the GPU box) → profiles/<round>_sq_summary.csv: per-kernel means
of the SQ / GRBM counters and the two fractions DESIGN.md quotes."""
import csv
import glob
import sys
from collections import defaultdict

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
acc = defaultdict(list)
for f in glob.glob("gpurun_out/lds/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("shk::", "")
        acc[(k, row["Counter_Name"])].append(float(row["Counter_Value"]))
cols = ["GRBM_GUI_ACTIVE", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS",
        "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
        "SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_ACTIVE_INST_ANY"]
with open(f"profiles/{rnd}_sq_summary.csv", "w") as out:
    out.write("# rocprofv3 --pmc passes (tools/pmc_lds.sh: two counters per pass with --kernel-trace only) over "
              "`python3 bench.py --steps 3 --warmup 1 --ramp-ms 0 --no-cpu-baseline --no-extras`; tools/sq_summary.py\n")
    out.write("# means per launch, summed over the device: GRBM_GUI_ACTIVE / 8 XCDs = cycles of the launch; a VALU "
              "instruction of a wave64 occupies its SIMD for 4 cycles (tools/ibench2.hip),\n")
    out.write("# so valu_busy_frac = SQ_INSTS_VALU * 4 / 1024 SIMDs / cycles; lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE\n")
    out.write("kernel," + ",".join(cols) + ",valu_busy_frac,lds_conflict_frac\n")
    for k in sorted({key[0] for key in acc}):
        if not (k.startswith("k_histo") or k.startswith("k_pages") or k.startswith("k_scatter")):
            continue
        m = {c: (sum(acc[(k, c)]) / len(acc[(k, c)]) if acc.get((k, c)) else 0.0) for c in cols}
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        vb = m["SQ_INSTS_VALU"] * 4 / 1024 / cyc if cyc else 0
        lc = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"] if m["SQ_LDS_IDX_ACTIVE"] else 0
        out.write(f'"{k}",' + ",".join(str(int(m[c])) for c in cols) + f",{vb:.3f},{lc:.3f}\n")
print(open(f"profiles/{rnd}_sq_summary.csv").read())

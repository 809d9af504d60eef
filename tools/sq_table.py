#!/usr/bin/env python3
"""tools/pmc_generic.sh's summary.txt (mean counter value per kernel) → a CSV under profiles/ with the derived columns the
analysis in DESIGN.md uses:
  cycles            = GRBM_GUI_ACTIVE / 8 XCDs (cycles of one launch)
  valu_busy_frac    = SQ_INSTS_VALU · 4 / 1024 SIMDs / cycles   (a wave64 VALU instruction occupies its SIMD for 4 cycles)
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  lds_busy_frac     = SQ_LDS_IDX_ACTIVE / 256 CUs / cycles
  waves_per_simd    = SQ_WAVE_CYCLES · 4 / cycles / 1024       (SQ_WAVE_CYCLES counts quad-cycles of resident waves)
  wait_inst_frac    = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES        (share of a resident wave's time spent stalled at issue)
  hbm_GB            = (2 · FETCH_SIZE + WRITE_SIZE) KiB → GB per launch (FETCH_SIZE doubled: gfx950 note of MI355X_MICROARCH.md)
usage: sq_table.py gpurun_out/<dir>/summary.txt profiles/<name>.csv "<what was run>" """
import re
import sys
from collections import defaultdict

src, dst, what = sys.argv[1], sys.argv[2], sys.argv[3]
acc = defaultdict(dict)
launches = {}
for line in open(src):
    m = re.match(r"(\S.*?)\s+(SQ_\w+|GRBM_\w+|FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+mean=\s*([\d.]+)", line)
    if m:
        acc[m.group(1).strip()][m.group(2)] = float(m.group(4))
        launches[m.group(1).strip()] = int(m.group(3))
cols = ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
        "SQ_INSTS_LDS", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD", "SQ_INSTS_SALU", "FETCH_SIZE", "WRITE_SIZE"]
with open(dst, "w") as out:
    out.write(f"# {what}\n# rocprofv3 --pmc passes (tools/pmc_generic.sh: separate passes, --kernel-trace only); means per launch, summed over the device; tools/sq_table.py\n")
    out.write("kernel,launches," + ",".join(cols) + ",cycles,valu_busy_frac,lds_busy_frac,lds_conflict_frac,waves_per_simd,wait_inst_frac,hbm_GB\n")
    for k in sorted(acc):
        m = acc[k]
        if not (k.startswith("k_p") or k.startswith("k_scatter") or k.startswith("k_histo") or k.startswith("k_xw") or k.startswith("k_direct")):
            continue
        g = lambda c: m.get(c, 0.0)
        cyc = g("GRBM_GUI_ACTIVE") / 8
        f = lambda x: f"{x:.3f}"
        out.write(f'"{k}",{launches[k]},' + ",".join(str(int(g(c))) for c in cols) + f",{int(cyc)}," +
                  ",".join([f(g("SQ_INSTS_VALU") * 4 / 1024 / cyc) if cyc else "", f(g("SQ_LDS_IDX_ACTIVE") / 256 / cyc) if cyc else "",
                            f(g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")) if g("SQ_LDS_IDX_ACTIVE") else "",
                            f(g("SQ_WAVE_CYCLES") * 4 / cyc / 1024) if cyc else "", f(g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")) if g("SQ_WAVE_CYCLES") else "",
                            f((2 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024 / 1e9)]) + "\n")
print(open(dst).read())

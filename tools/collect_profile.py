#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the
committed evidence under profiles/: the rocprofv3 --stats kernel summary, the PMC counter
means per kernel, and <round>_traffic.json (HBM bytes per launch per kernel; FETCH_SIZE is
doubled, the gfx950 correction of MI355X_MICROARCH.md §HBM — WRITE_SIZE is exact)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
reads = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
k = int(sys.argv[4]) if len(sys.argv) > 4 else 21
batches = int(sys.argv[5]) if len(sys.argv) > 5 else 4   # bench.py rotates its steps over this many distinct resident batches
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, f"profiles/{rnd}_kernel_stats.csv")

# kernel-name prefix → short name used by bench.py (first match wins)
NAMES = [("k_scatter32", "scatter"), ("k_scatter64", "scatter"), ("k_part_scatter_sorted", "scatter"), ("k_pages32", "pages"), ("k_hist_reduce", "histo_rows"), ("k_ctl_out", "ctl_out"),
         ("k_pages", "pages"), ("k_part_rescatter", "rescatter"), ("k_fill", "fill"), ("k_histo", "histo"),
         ("k_direct", "direct"), ("k_scan", "scan"), ("k_mark_starts", "mark")]


def means(sub):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("shk::", "")
            acc[(kn, row["Counter_Name"])].append(float(row["Counter_Value"]))
    return {key: sum(v) / len(v) for key, v in acc.items()}


fetch, write = means("pmc_fetch"), means("pmc_write")
with open(f"profiles/{rnd}_pmc_summary.csv", "w") as f:
    f.write("kernel,counter,mean_per_launch_KB\n")
    for (kn, c), v in sorted({**fetch, **write}.items()):
        f.write(f"{kn},{c},{v:.1f}\n")
kernels = {}
for kn in sorted({key[0] for key in list(fetch) + list(write)}):
    short = next((sh for pre, sh in NAMES if kn.startswith(pre)), None)
    if short is None or short in kernels:
        continue
    fs, ws = fetch.get((kn, "FETCH_SIZE")), write.get((kn, "WRITE_SIZE"))
    fb, wb = 2 * (fs or 0) * 1024, (ws or 0) * 1024
    kernels[short] = {"kernel": kn, "fetch_bytes_corrected": int(fb), "write_bytes": int(wb),
                      "hbm_bytes_per_launch": int(fb + wb)}
json.dump({"reads": reads, "k": k, "batches": batches, "source": f"tools/profile_round.sh {tag}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
           "in separate passes over `python3 bench.py --no-cpu-baseline --no-extras` (the default run: steps rotate over "
           f"{batches} distinct batches)",
           "kernels": kernels}, open(f"profiles/{rnd}_traffic.json", "w"), indent=1)
print(json.dumps(kernels, indent=1))

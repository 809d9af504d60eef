#!/bin/bash
# rocprofv3 evidence for `bench.py --config 4 / 5` on one card (one rank's share of BASELINE configs[3] / configs[4]
# through the exchange rounds): kernel stats, then the FETCH_SIZE and WRITE_SIZE passes on their own (separate --pmc
# passes, --kernel-trace only).  usage: bash tools/profile_owner.sh <4|5> <tag>; outputs under gpurun_out/<tag>/, the
# summaries to copy into profiles/: <tag>/kernel_stats.csv, <tag>/traffic.json (HBM bytes per launch per kernel).
set -e
CFG=$1; TAG=${2:-r04c$1}
ARGS="--config $CFG --no-cpu-baseline --no-extras --steps 2 --warmup 1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 bench.py $ARGS > gpurun_out/$TAG/trace.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_fetch -- python3 bench.py $ARGS > gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_write -- python3 bench.py $ARGS > gpurun_out/$TAG/pmc_write.log 2>&1
cp $(find gpurun_out/$TAG/trace -name '*kernel_stats.csv' | head -1) gpurun_out/$TAG/kernel_stats.csv
find gpurun_out/$TAG -name '*kernel_trace.csv' -size +8M -delete
python3 - <<PY
import csv, glob, json
from collections import defaultdict
src = "gpurun_out/$TAG"
NAMES = [("k_scatter32", "scatter"), ("k_part_scatter_sorted", "scatter"), ("k_pages32", "pages"), ("k_pages", "pages"),
         ("k_part_rescatter32", "pscan"), ("k_part_rescatter", "pscan"), ("k_histo", "histo"), ("k_mark_starts", "mark")]
def means(sub):
    acc = defaultdict(list)
    for f in glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("shk::", "")
            acc[kn].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fe, wr = means("pmc_fetch"), means("pmc_write")
kernels = {}
# a short name's figure is the launch-weighted mean over the kernels that carry it (the fresh and the reading page pass
# of one job are two instantiations of k_pages32; bench.py's per-kernel time averages over both the same way)
for short in sorted(set(sh for _, sh in NAMES)):
    def of(kn):
        return next((sh for pre, sh in NAMES if kn.startswith(pre)), None) == short
    kns = sorted(kn for kn in set(fe) | set(wr) if of(kn))
    if not kns:
        continue
    n = sum(max(fe.get(kn, (0, 0))[1], wr.get(kn, (0, 0))[1]) for kn in kns)
    fb = sum(2 * fe.get(kn, (0, 0))[0] * 1024 * fe.get(kn, (0, 0))[1] for kn in kns) / n
    wb = sum(wr.get(kn, (0, 0))[0] * 1024 * wr.get(kn, (0, 0))[1] for kn in kns) / n
    kernels[short] = {"kernel": " + ".join(kns), "launches_seen": n,
                      "fetch_bytes_corrected": int(fb), "write_bytes": int(wb), "hbm_bytes_per_launch": int(fb + wb)}
json.dump({"config": $CFG, "command": "python3 bench.py $ARGS",
           "source": "tools/profile_owner.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; means per launch; FETCH_SIZE doubled (gfx950 note, MI355X_MICROARCH.md §HBM)",
           "kernels": kernels}, open(f"{src}/traffic.json", "w"), indent=1)
print(json.dumps(kernels)[:1200])
PY

#!/bin/bash
# rocprofv3 evidence for BASELINE configs[2] at full size from HBM (tools/config3_run.py): kernel stats, then the
# FETCH_SIZE and WRITE_SIZE passes on their own.  Outputs under gpurun_out/<tag>/.
set -e
TAG=${1:-r03c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/trace -- python3 tools/config3_run.py > gpurun_out/$TAG/trace.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_fetch -- python3 tools/config3_run.py > gpurun_out/$TAG/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_write -- python3 tools/config3_run.py > gpurun_out/$TAG/pmc_write.log 2>&1
find gpurun_out/$TAG -name '*kernel_trace.csv' -size +8M -delete
python3 - <<PY
import csv, glob, json
from collections import defaultdict
src = "gpurun_out/$TAG"
def sums(sub):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            kn = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("shk::", "").split("<")[0]
            a = acc[kn]; a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc
fe, wr = sums("pmc_fetch"), sums("pmc_write")
out = {"command": "python3 tools/config3_run.py (100 M reads, k = 31, 300 Mb genome, from HBM; generation + 1 counted pass)",
       "note": "FETCH_SIZE doubled (gfx950: counts 32-B units as 64-B ones… see MI355X_MICROARCH.md, HBM section); WRITE_SIZE as read; KB -> bytes; sums over ALL launches of the run",
       "kernels": {}}
for kn in sorted(set(fe) | set(wr)):
    out["kernels"][kn] = {"launches": max(fe.get(kn, [0, 0])[1], wr.get(kn, [0, 0])[1]),
                          "fetch_GB": round(fe.get(kn, [0, 0])[0] * 2 * 1024 / 1e9, 3), "write_GB": round(wr.get(kn, [0, 0])[0] * 1024 / 1e9, 3)}
json.dump(out, open(f"{src}/traffic.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY

#!/bin/bash
# Experiment: does the fast/slow mode of a process correlate with address-translation misses?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tlb
for i in 1 2 3 4 5 6; do
  timeout -k 10 200 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d gpurun_out/tlb/p$i -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --ramp-ms 0 > gpurun_out/tlb/p$i.log 2>&1 || echo "pass $i failed"
  python3 tools/pmcsum.py gpurun_out/tlb/p$i | grep -E "k_pages|k_part" 
  python3 - <<PY
import csv,glob
for f in glob.glob("gpurun_out/tlb/p$i/**/*kernel_trace.csv", recursive=True):
    d={}
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"].split("(")[0].replace("void ","").replace("shk::","")
        d.setdefault(n,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000)
    for n,v in d.items():
        if n.startswith("k_pa"): print("   dur", n[:24], round(sum(v)/len(v),1))
PY
done

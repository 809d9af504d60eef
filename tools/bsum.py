#!/usr/bin/env python3
"""Print a compact summary of bench.py JSON lines read from stdin (experiment helper)."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    r = d.get("roofline") or {}
    print(tag, "Gb/s", d["value"], "ms", d["ms_per_step"], "frac", r.get("frac"),
          d.get("kernels_ms_per_step"), flush=True)

#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for kb in 65536 98304 131072 262144 393216; do
  echo "== SHK_SLICE_KB=$kb"
  SHK_SLICE_KB=$kb python3 tools/config3_host_probe.py 24000000 4000000 2>/dev/null | tail -1
done
echo "== default"; python3 tools/config3_host_probe.py 24000000 4000000 2>/dev/null | tail -1

#!/bin/bash
# one large .fastq.gz through the front-end alone, with the decoder's own timing lines
cd "$GRAFT_REPO_ROOT"
g++ -O2 -o /tmp/reader_bench tools/reader_bench.cpp -Lsharkmer_amd/csrc -lshk -Wl,-rpath,$PWD/sharkmer_amd/csrc || exit 1
python3 - <<'PY'
import numpy as np, gzip, shutil
n, L = 4_000_000, 150
rng = np.random.default_rng(1)
rec = np.empty((n, 2 * L + 7), dtype=np.uint8)
rec[:, 0:3] = np.frombuffer(b"@r\n", dtype=np.uint8)
rec[:, 3:3 + L] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=(n, L))]
rec[:, 3 + L:6 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
rec[:, 6 + L:6 + 2 * L] = ord("I")
rec[:, 6 + 2 * L] = ord("\n")
rec.tofile("/tmp/big.fastq")
with open('/tmp/big.fastq','rb') as f, gzip.open('/tmp/big.fastq.gz','wb',compresslevel=1) as g: shutil.copyfileobj(f,g,1<<24)
PY
ls -la /tmp/big.fastq /tmp/big.fastq.gz
/tmp/reader_bench --packed /tmp/big.fastq.gz | tail -3
SHK_FASTQ_DEBUG=1 /tmp/reader_bench --packed /tmp/big.fastq.gz 2>&1 | grep -i "gzip member\|workers" | tail -4
echo "== plain"; /tmp/reader_bench --packed /tmp/big.fastq | tail -2

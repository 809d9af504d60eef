import sys, numpy as np
sys.path.insert(0, '.')
import torch
import sharkmer_amd as sa
L=150; n=1_000_000
spec = sa.SynthSpec(genome_len=3_000_000, read_len=L)
eng = sa.KmerEngine(21, 1, 10000, capacity_hint=3_000_000)
d_b = torch.empty(n*L, dtype=torch.uint8, device="cuda:0"); d_o = torch.empty(n+1, dtype=torch.int64, device="cuda:0")
eng.synth_reads_device(spec, 0, n, d_b.data_ptr(), d_o.data_ptr())
eng.ingest_reads_device(d_b.data_ptr(), d_o.data_ptr(), n, n*L)
eng.finalize()
k, c = eng.export_table()
M = np.uint64(0xC2B2AE35); mask = np.uint64((1<<42)-1)
y = (k * M) & mask
page = (y >> np.uint64(32)).astype(np.int64); bucket = ((y >> np.uint64(21)) & np.uint64(2047)).astype(np.int64)
sel = (page == 118) & (bucket >= 770) & (bucket <= 795)
print("keys in page 118:", int((page==118).sum()), " in buckets 770..795:", int(sel.sum()))
def seq(x):
    return "".join("ACGT"[(int(x) >> (2*(20-i))) & 3] for i in range(21))
for kk, bb, cc in sorted(zip(k[sel].tolist(), bucket[sel].tolist(), c[sel].tolist()), key=lambda t: t[1]):
    print(bb, hex(kk), seq(kk), cc)
bc = np.bincount(page*2048+bucket, minlength=1024*2048)
print("max bucket load", bc.max(), "buckets with >= 8 keys", int((bc>=8).sum()))

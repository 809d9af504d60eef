import os, sys, time, json
# Per-round host times of the exchange scatter and absorb calls (what found the runtime's one-time stall).
# The probe times the ONE-call scatter (xchg_scatter_tensors), so it asks OwnerCounter for that form.
os.environ.setdefault("SHK_DIST_ONE_CALL_SCATTER", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
import sharkmer_amd as sa
from sharkmer_amd.dist import OwnerCounter
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29577")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
reads, genome, L, k = int(os.environ.get("PROBE_READS", "62500000")), 3_000_000_000, 150, 21
spec = sa.SynthSpec(genome_len=genome, read_len=L)
rr = 1_700_000; n_rounds = -(-reads // rr)
eng = sa.KmerEngine(k, 1, 1000, capacity_hint=genome, flags=(sa.FLAG_TIMING if os.environ.get("PROBE_TIMING", "1") == "1" else 0), n_owners=1)
d_all = torch.empty(reads*L, dtype=torch.uint8, device="cuda:0"); d_off = torch.empty(rr+1, dtype=torch.int64, device="cuda:0")
for r in range(n_rounds):
    n = min(rr, reads - r*rr); eng.synth_reads_device(spec, r*rr, n, d_all.data_ptr()+r*rr*L, d_off.data_ptr())
eng.sync()
oc = OwnerCounter(eng, dist, device=0, round_bases=rr*L)
acc = {"absorb":0.0, "scatter":0.0, "n":0}
orig_abs, orig_sc = eng.xchg_absorb_tensors, eng.xchg_scatter_tensors
def wa(*a, **k2):
    t=time.perf_counter(); r=orig_abs(*a, **k2); acc["absorb"] += time.perf_counter()-t; acc["n"]+=1; return r
def ws(*a, **k2):
    t=time.perf_counter(); r=orig_sc(*a, **k2); acc["scatter"] += time.perf_counter()-t; return r
eng.xchg_absorb_tensors, eng.xchg_scatter_tensors = wa, ws
for rep in range(3):
    eng.reset(); acc.update(absorb=0.0, scatter=0.0, n=0)
    torch.cuda.synchronize(); t0=time.time()
    per = []
    for r in range(n_rounds):
        n = min(rr, reads - r*rr)
        a0, s0, tr0 = acc["absorb"], acc["scatter"], time.perf_counter()
        oc.round((d_all.data_ptr()+r*rr*L, d_off.data_ptr(), n, n*L, r*rr))
        per.append((round((time.perf_counter()-tr0)*1e3,2), r, round((acc["scatter"]-s0)*1e3,2), round((acc["absorb"]-a0)*1e3,2)))
    t1=time.time(); oc.finalize_histograms(); torch.cuda.synchronize(); t2=time.time()
    print("slowest rounds (ms, round, scatter call, absorb call):", sorted(per, reverse=True)[:4], flush=True)
    print(json.dumps({"rep":rep, "rounds_s": round(t1-t0,4), "finalize_s": round(t2-t1,4), "absorb_ms_per_call": round(acc["absorb"]/max(acc["n"],1)*1e3,3), "scatter_ms_per_call": round(acc["scatter"]/n_rounds*1e3,3)}), flush=True)
eng.close(); dist.destroy_process_group()

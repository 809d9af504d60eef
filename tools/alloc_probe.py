import sys, os, time
sys.path.insert(0, os.getcwd())
import sharkmer_amd as sa
eng = sa.KmerEngine(21, 1, 100)
for gb in (1, 4, 10, 14, 10):
    t0 = time.time(); p = eng.alloc_device(gb << 30); t1 = time.time(); eng.free_device(p); t2 = time.time()
    print(gb, "GiB malloc %.3f s free %.3f s" % (t1 - t0, t2 - t1), flush=True)

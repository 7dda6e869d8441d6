"""Per-rank kernel times of the decomposed 1M-atom box with `world` virtual ranks on one GPU
(ThreadComm).  The ranks share the GPU, so wall time is meaningless; the HIP-event kernel times and
the owned/ghost counts tell what each rank of a real multi-GPU run has to do per step."""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.decomp import DecomposedMd, ThreadComm

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 70
s = systems.water_box(n_side)
cfg = MdConfig()
shared = ThreadComm.Shared(world)
res, errs = {}, []

def run(rank):
    try:
        md = DecomposedMd(s, cfg, rank=rank, world=world, device=0, comm=ThreadComm(rank, shared))
        md.step(0.0005, 10)
        md.profile(True)
        t = time.time(); md.step(0.0005, 40); el = time.time() - t
        res[rank] = (md.stats(), el)
    except BaseException as e:
        errs.append(e); shared.barrier.abort()

th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
[t.start() for t in th]; [t.join() for t in th]
if errs: raise errs[0]
for r in sorted(res):
    st, el = res[r]
    print("rank %d: owned %d ghost %d tiles %d entries %d | nb %.3f ms bonded %.3f integ %.3f | repartitions %d (%.2f ms each, host wall) rebuild_ms %.2f | wall %.2fs" % (
        r, st["n_owned"], st["n_ghost"], st["n_tiles"], st["n_list_entries"], st["nb_ms_sum"]/max(st["nb_launches"],1),
        st["bonded_ms_sum"]/max(st["bonded_launches"],1), st["integ_ms_sum"]/max(st["integ_launches"],1), st["repartitions"], st["repartition_ms_sum"]/max(st["repartitions"]-1,1), st["rebuild_ms_sum"], el))

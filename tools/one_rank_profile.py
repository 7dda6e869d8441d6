"""What ONE rank of an N-GPU run has to do per step, measured alone on the GPU: rank 0 of the N-rank decomposition of the
1M-atom box with a communicator that delivers nothing (ghosts keep their positions, reductions are the identity).  The
kernel and host costs per step are those of a real rank; only the wire time of the halo message is missing.
Usage: timeout 200 python tools/one_rank_profile.py [world] [steps=60]   (keep the timeout: a wrong repartition of a box
whose other ranks do not exist piles a million atoms on one point)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from molchanica_amd import systems, MdConfig
from molchanica_amd.decomp import DecomposedMd


class FrozenComm:
    def __init__(self, world): self.rank, self.world = 0, world
    def all_reduce(self, t, op): pass
    def prepare(self, sends, recvs): return None
    def prepare_halo(self, send_buf, recv_buf, send_segs, recv_segs): return None
    def run(self, prepared): pass
    def exchange(self, sends, recvs): pass


worlds = [int(sys.argv[1])] if len(sys.argv) > 1 else [1, 2, 4, 8]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
s = systems.water1m() if "ONE_RANK_NSIDE" not in os.environ else systems.water_box(int(os.environ["ONE_RANK_NSIDE"]))   # a small box shows the host-side floor per step
if os.environ.get("ONE_RANK_HOT", "0") != "1":      # same untimed preparation as bench.py: relaxed, 300 K
    import numpy as np
    from molchanica_amd.md_state import MdState
    with MdState(s, MdConfig()) as eq:
        eq.minimize_energy(100); eq.initialize_velocities(300.0, True, seed=105)
        eq.set_thermostat(1, 300.0, 0.02, 1); eq.step(0.0005, None, 600); eq.set_thermostat(0, 300.0, 0.02, 1)
        s.pos = np.ascontiguousarray(eq.positions(), dtype=np.float32); s.vel = np.ascontiguousarray(eq.velocities(), dtype=np.float32)
for world in worlds:
    md = DecomposedMd(s, MdConfig(), rank=0, world=world, device=0, comm=FrozenComm(world))
    md._local_set_still_valid = lambda: True      # never repartition: the other ranks' rows do not exist here
    md.recv_ids = md.recv_ids[:0]                 # nothing arrives: skip the unpack (a ~4 us kernel), ghosts stay put
    md.step(0.0005, 8)
    torch.cuda.synchronize()
    md.profile(1)
    t0 = time.perf_counter(); md.step(0.0005, steps); torch.cuda.synchronize(); el = time.perf_counter() - t0
    st = md.stats()
    print("world %d rank 0: owned %d ghost %d tiles %d | pair %.3f ms bonded %.3f integrate %.3f | step wall %.3f ms "
          "(%d list rebuilds in %d steps, %.2f ms each) -> ceiling %.0f steps/s without wire time | cluster pairs verlet %.1f M inner %.1f M" % (
              world, st["n_owned"], st["n_ghost"], st["n_tiles"], st["nb_ms_sum"] / max(st["nb_launches"], 1),
              st["bonded_ms_sum"] / max(st["bonded_launches"], 1), st["integ_ms_sum"] / max(st["integ_launches"], 1),
              1e3 * el / steps, st["rebuild_count"], steps + 8, st["rebuild_ms_sum"] / max(st["rebuild_count"] - 1, 1), steps / el, st["n_cluster_pairs"] / 1e6, st["n_inner_cluster_pairs"] / 1e6), flush=True)
    del md

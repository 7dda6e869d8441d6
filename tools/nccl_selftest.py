"""Single-rank RCCL plumbing check on the 1-GPU box: process group on backend nccl, collectives and
a batched self send/recv issued under the engine's external HIP stream, mixed with engine kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.decomp import HipEngine, DistComm

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = systems.water_box(8)
cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5)
eng = HipEngine(s, cfg, 0)
comm = DistComm(0, 1)
with torch.cuda.stream(eng.stream):
    eng.chunk_begin(); eng.chunk_forces(-1)
    eng.chunk_integrate(0, 0.0005, 0)
    flags = eng.flag_tensor()
    tmp = torch.zeros(1, dtype=torch.int32, device="cuda")
    tmp.copy_(flags[1:2]); comm.all_reduce(tmp, "max"); flags[1:2].copy_(tmp)
    st = torch.arange(12, dtype=torch.float32, device="cuda").reshape(3, 4)
    comm.all_reduce(st, "sum")
    gid = torch.tensor([0, 5, 9, -1], dtype=torch.int32, device="cuda")
    sbuf = torch.zeros((4, 4), device="cuda"); rbuf = torch.zeros((4, 4), device="cuda")
    eng.pack(gid, sbuf, 1)
    try:
        comm.exchange([(0, sbuf)], [(0, rbuf)])
        ok_self = bool(torch.equal(sbuf, rbuf))
    except Exception as e:
        ok_self = f"self send/recv unsupported: {type(e).__name__}: {str(e)[:80]}"
    try:
        rbuf2 = torch.zeros((4, 4), device="cuda")
        comm.run(comm.prepare_halo(sbuf, rbuf2, [(0, 0, 4)], [(0, 0, 4)]))      # the default halo path: all_to_all_single
        ok_a2a = bool(torch.equal(sbuf, rbuf2))
    except Exception as e:
        ok_a2a = f"all_to_all_single failed: {type(e).__name__}: {str(e)[:80]}"
    eng.chunk_forces(0)
    eng.chunk_integrate(2, 0.0005, 1)
    words = eng.chunk_end(2)
torch.cuda.synchronize()
print("all_reduce ok:", bool(torch.equal(st.cpu(), torch.arange(12.).reshape(3, 4))), "| flag word:", int(words[1]),
      "| self exchange:", ok_self, "| all_to_all_single halo:", ok_a2a, "| packed row0:", sbuf[0].tolist())
dist.destroy_process_group()

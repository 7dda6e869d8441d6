"""Cost of a repartition of the decomposed driver on one GPU (world = 1: no communication, only the host + torch work
and the engine's set_local_atoms + rebuild)."""
import sys, time, torch
sys.path.insert(0, ".")
from molchanica_amd import systems, MdConfig
from molchanica_amd.decomp import DecomposedMd
torch.cuda.set_device(0)
s = systems.water1m()
md = DecomposedMd(s, MdConfig(), rank=0, world=1, device=0)
md.step(0.0005, 20)
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    with md._stream():
        pos, vel = md._gather_global()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        md._repartition_from(pos, vel)
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"gather_global {1e3*(t1-t0):.2f} ms   repartition_from {1e3*(t2-t1):.2f} ms")
md.step(0.0005, 5)
print("ok", md.step_count)

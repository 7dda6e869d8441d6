"""Where the decomposed driver's host time goes (world = 1, dna100k): cProfile of 400 steps."""
import sys, time, cProfile, pstats, torch
sys.path.insert(0, ".")
from molchanica_amd import systems, MdConfig
from molchanica_amd.decomp import DecomposedMd
torch.cuda.set_device(0)
s = systems.dna100k()
md = DecomposedMd(s, MdConfig(), rank=0, world=1, device=0)
md.step(0.0005, 50); torch.cuda.synchronize()
t0 = time.perf_counter(); md.step(0.0005, 400); torch.cuda.synchronize(); print("ms/step", (time.perf_counter() - t0) / 400 * 1e3, "repartitions", md.repartitions, "repartition_s", md.repartition_s)
pr = cProfile.Profile(); pr.enable(); md.step(0.0005, 400); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

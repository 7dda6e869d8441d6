import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, ctypes as C
from molchanica_amd import systems, MdConfig
from molchanica_amd import md_state as M
s = systems.BY_NAME["complex50k"](); cfg = MdConfig()
lig = slice(int(s.mol_start[1]), int(s.mol_start[2])); rng = np.random.default_rng(1); base = s.pos.copy()
M.release_single_point_cache(); M.compute_energy_snapshot(s, cfg)
lib = M.load_library()
T = dict(copy=0.0, norm=0.0, toc=0.0, call=0.0); n = 60
for k in range(n):
    t0 = time.perf_counter(); p = base.copy(); p[lig] += rng.normal(0, 0.15, 3).astype(np.float32); s.pos = p
    t1 = time.perf_counter(); s.normalise()
    t2 = time.perf_counter(); cs, cc = s.to_c(), cfg.to_c(); e = M.CEnergies()
    t3 = time.perf_counter(); rc = lib.mdx_single_point(C.byref(cs), C.byref(cc), 0, C.byref(e), None)
    t4 = time.perf_counter()
    if k >= 10:
        T["copy"] += t1 - t0; T["norm"] += t2 - t1; T["toc"] += t3 - t2; T["call"] += t4 - t3
print({k: round(1e6 * v / (n - 10), 1) for k, v in T.items()}, "us per pose; ligand atoms", lig)

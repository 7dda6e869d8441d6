"""Diagnostic: decomposed rigid-water run vs single GPU; where do the deviations sit?  (gpurun -- python tools/dd_diag.py)"""
import sys, os, math, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import MdConfig, systems
from molchanica_amd.md_state import MdState, Fabric

def run_ranks(system, cfg, world, n_steps, dt, setup=None):
    fabric = Fabric(world); res, errs = {}, []
    def run(rank):
        try:
            with MdState(system, cfg) as md:
                if setup: setup(md)
                md.comm_init_fabric(fabric, rank)
                md.step(dt, None, n_steps)
                res[rank] = dict(pos=md.positions(), stats=md.stats(), e=md.energy())
        except BaseException as e:
            errs.append(e); fabric.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    if errs: raise errs[0]
    return res

s = systems.opc_water_box(16, seed=3)
L = np.array(s.box_hi, dtype=np.float64)
for label, thermo, dt, nsteps, inner in [("csvr 2fs", True, 0.002, 40, 0.0), ("nve 2fs", False, 0.002, 40, 0.0), ("nve 2fs plain list", False, 0.002, 40, -1.0),
                                          ("nve 2fs 8 steps", False, 0.002, 8, 0.0), ("nve 0.5fs", False, 0.0005, 40, 0.0)]:
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=1, chunk_steps=8, inner_skin=inner)
    setup = (lambda md: md.set_thermostat(2, 300.0, 0.1, 5, seed=77)) if thermo else None
    with MdState(s, cfg) as md:
        if setup: setup(md)
        md.step(dt, None, nsteps)
        p_ref = md.positions().astype(np.float64); st_ref = md.stats(); e_ref = md.energy()
    for world, env in [(2, {}), (2, {"MDX_HALO_OVERLAP": "0"})]:
        os.environ.pop("MDX_HALO_OVERLAP", None); os.environ.update(env)
        res = run_ranks(s, cfg, world, nsteps, dt, setup)
        d = res[0]["pos"].astype(np.float64) - p_ref
        d -= np.round(d / L) * L
        n = np.linalg.norm(d, axis=1)
        bad = np.nonzero(n > 1e-2)[0]
        print(f"{label:20s} world {world} {env}: rms {math.sqrt((n**2).mean()):.2e} max {n.max():.2e} n_bad {len(bad)} "
              f"rebuilds ref {st_ref['rebuild_count']} dd {res[0]['stats']['rebuild_count']} repart {res[0]['stats']['repartitions']} "
              f"T ref {e_ref['temperature']:.2f} dd {res[0]['e']['temperature']:.2f}")
        if len(bad):
            x = p_ref[bad]
            print("   bad atoms: kinds", np.bincount(bad % 4, minlength=4), " x range", x[:, 0].min(), x[:, 0].max(),
                  " dist to x-faces (0, L/2):", np.minimum(np.abs(x[:, 0] - 0), np.minimum(np.abs(x[:, 0] - L[0] / 2), np.abs(x[:, 0] - L[0]))).max())

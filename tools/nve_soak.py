"""Energy conservation from EQUILIBRATED, wrapped states - the states a running box is in, which the generators' lattices
of whole molecules are not (round 2 found virtual sites rebuilt in the wrong periodic image only this way).  For every
system / configuration: minimise, thermalise 3 ps under a Berendsen thermostat, switch it off, run NVE and print the
drift of the total energy per 1000 steps relative to the kinetic energy.
Usage (through gpurun): python tools/nve_soak.py > gpurun_out/nve_soak.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState


def soak(name, s, cfg, dt, n_eq=3000, n_nve=2000, eq_dt=None):
    t0 = time.perf_counter()
    with MdState(s, cfg) as md:
        md.minimize_energy(200)
        md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.05, 1)
        md.step(eq_dt or min(dt, 0.001), None, n_eq)
        md.step(dt, None, 500)
        md.set_thermostat(0, 300.0, 0.05, 1)
        e = [md.energy()]
        for _ in range(4):
            md.step(dt, None, n_nve // 4)
            e.append(md.energy())
        st = md.stats()
    tot = np.array([x["potential"] + x["kinetic"] for x in e])
    drift = (tot[-1] - tot[0]) * 1000.0 / n_nve
    print(f"{name:58s} dt {dt * 1e3:4.2f} fs  T {e[0]['temperature']:6.1f} -> {e[-1]['temperature']:6.1f} K  "
          f"E_tot {tot[0]:12.1f}  drift {drift:9.2f} kcal/mol per 1000 steps = {100 * drift / e[0]['kinetic']:7.3f} % of E_kin  "
          f"(spread {tot.max() - tot.min():7.2f}; {st['rebuild_count']} rebuilds; {time.perf_counter() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    rf, cut = dict(coulomb_mode=1), dict()
    spme = dict(coulomb_mode=2, ewald_alpha=0.3, overrides=0)
    soak("flexible TIP3P 17 k, shifted cutoff", systems.water_box(18, seed=5), MdConfig(**cut), 0.0005)
    soak("flexible TIP3P 17 k, reaction field", systems.water_box(18, seed=5), MdConfig(**rf), 0.0005)
    soak("flexible TIP3P 17 k, SPME", systems.water_box(18, seed=5), MdConfig(**spme), 0.0005)
    soak("rigid TIP3P 17 k, reaction field", systems.water_box(18, seed=5, rigid=True), MdConfig(**rf), 0.002)
    soak("rigid TIP3P 17 k, SPME", systems.water_box(18, seed=5, rigid=True), MdConfig(**spme), 0.002)
    soak("rigid OPC 23 k sites, reaction field", systems.opc_water_box(18, seed=5), MdConfig(**rf), 0.002)
    soak("rigid OPC 23 k sites, SPME (the reference's default point)", systems.opc_water_box(18, seed=5), MdConfig(**spme), 0.002)
    soak("rigid OPC 23 k sites, SPME, full list (nb_variant 2)", systems.opc_water_box(18, seed=5), MdConfig(nb_variant=2, **spme), 0.002)
    soak("dhfr23k (solvated chain, flexible), shifted cutoff", systems.BY_NAME["dhfr23k"](), MdConfig(**cut), 0.0005)
    soak("dhfr23k, SPME", systems.BY_NAME["dhfr23k"](), MdConfig(**spme), 0.0005)
    soak("dna100k, reaction field", systems.BY_NAME["dna100k"](), MdConfig(**rf), 0.0005, n_eq=2000, n_nve=1000)

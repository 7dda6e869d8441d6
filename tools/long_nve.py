"""100 ps of NVE at the reference's default operating point (rigid OPC, SPME, dt 2 fs), 20 ps of the solvated chain, 10 ps of the
1 M-atom box, each after a CSVR-thermostatted lead-in: total-energy drift per atom and nanosecond.  Usage (through gpurun):
python tools/long_nve.py > gpurun_out/long_runs.txt   (profiles/r02_long_runs.txt)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
def run(name, s, cfg, dt, n_eq, n_nve, thermo_in_eq=True):
    t0 = time.perf_counter()
    with MdState(s, cfg) as md:
        md.minimize_energy(200); md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.05, 1); md.step(min(dt, 0.001), None, 3000)
        md.set_thermostat(2, 300.0, 0.1, 10, seed=2); md.step(dt, None, n_eq)
        e_nvt = md.energy()
        md.set_thermostat(0, 300.0, 0.1, 10)
        e = [md.energy()]
        for k in range(5):
            md.step(dt, None, n_nve // 5); e.append(md.energy())
        st = md.stats()
    tot = np.array([x["potential"] + x["kinetic"] for x in e])
    ns = n_nve * dt * 1e-3
    print(f"{name}: {n_eq} steps NVT (CSVR) -> T {e_nvt['temperature']:.1f} K; then {n_nve} steps NVE = {ns * 1e3:.0f} ps: T {e[0]['temperature']:.1f} -> {e[-1]['temperature']:.1f} K, "
          f"E_tot drift {tot[-1] - tot[0]:.1f} kcal/mol = {100 * (tot[-1] - tot[0]) / e[0]['kinetic']:.3f} % of E_kin ({(tot[-1] - tot[0]) / s.n_atoms / ns:.4f} kcal/mol/atom/ns), "
          f"{st['rebuild_count']} list rebuilds, wall {time.perf_counter() - t0:.0f} s", flush=True)
run("rigid OPC 23,328 sites, SPME, dt 2 fs (the reference's default operating point)", systems.opc_water_box(18, seed=5), MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0), 0.002, 25000, 50000)
run("dhfr23k flexible chain + water, SPME, dt 0.5 fs", systems.BY_NAME["dhfr23k"](), MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0), 0.0005, 10000, 40000)
run("water1M flexible TIP3P, reaction field, dt 0.5 fs", systems.BY_NAME["water1M"](), MdConfig(coulomb_mode=1), 0.0005, 2000, 20000)
# round 4: the paths only the large default-point box takes (brick charge spread, clusters by interaction kind, cluster table in slot order)
run("rigid OPC 1,048,576 sites, SPME, dt 2 fs", systems.opc_water_box(64, seed=5), MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0), 0.002, 1500, 5000)

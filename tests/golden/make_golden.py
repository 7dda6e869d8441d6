"""Generates the committed golden vectors from the CPU oracle (NOT from the reference: its MD
engine is an external crate that cannot be built here — SURVEY.md §8c).  Re-run only when the
oracle's documented conventions change:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molchanica_amd import MdConfig, systems  # noqa: E402
from oracle import oracle  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # S0/C1: 50-atom ligand, vacuum, no cutoff
    "lig50": (lambda: systems.lig50(seed=1), dict(lj_cutoff=0.0, coulomb_cutoff=0.0)),
    # 216 flexible TIP3P waters, periodic, rc 7 Å + 1.5 Å skin
    "water648": (lambda: systems.water_box(6, seed=5), dict(lj_cutoff=7.0, coulomb_cutoff=7.0, skin=1.5)),
}


def main():
    for name, (mk, kw) in CASES.items():
        s = mk()
        cfg = MdConfig(**kw)
        f, e = oracle.forces(s, cfg)
        out = dict(pos=s.pos, forces=f, **{f"e_{k}": np.float64(v) for k, v in e.items()})
        if s.periodic:
            off, idx = oracle.neighbor_list(s, cfg.lj_cutoff + cfg.skin)
            out.update(nl_offsets=off, nl_idx=idx)
        x, v, e2 = oracle.step(s, cfg, 0.0005, 20)
        out.update(pos20=x, vel20=v, e20_potential=np.float64(e2["potential"]), e20_kinetic=np.float64(e2["kinetic"]))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, s.n_atoms, {k: round(v, 4) for k, v in e.items()})


if __name__ == "__main__":
    main()

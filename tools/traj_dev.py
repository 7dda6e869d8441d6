import numpy as np, math, sys
sys.path.insert(0,'/root/repo')
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
from oracle import oracle as orc
s = systems.small_solvated(n_chain=240, box=30.0)
for mode in (1,0):
    cfg = MdConfig(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, coulomb_mode=mode)
    with MdState(s, cfg) as md:
        x0, v0 = md.positions().astype(np.float64), md.velocities().astype(np.float64)
        outs={}
        for n in (25,50,100):
            md.step(0.0005, None, n - md.step_count)
            outs[n]=(md.positions().astype(np.float64), md.velocities().astype(np.float64))
    L = np.array(s.box_hi) - np.array(s.box_lo)
    for n in (25,50,100):
        xo, vo, _ = orc.step(s, cfg, 0.0005, n, pos=x0, vel=v0, use_cells=True)
        xg,vg=outs[n]
        d = xg - xo; d -= np.round(d / L) * L
        dv=vg-vo
        print(mode, n, "pos rms %.2e max %.2e | vel rms %.3e max %.3e argmax %d" % (math.sqrt((d**2).sum(1).mean()), np.abs(d).max(), math.sqrt((dv**2).sum(1).mean()), np.abs(dv).max(), np.argmax(np.abs(dv).max(1))))

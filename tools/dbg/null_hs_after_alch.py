import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import pytest
pytest.main(["tests/test_gpu_alchemical.py", "-m", "gpu", "-q", "-x"] + sys.argv[1:])
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState
CFG = dict(lj_cutoff=9.0, coulomb_cutoff=9.0, skin=1.5, chunk_steps=8)
for wire, pin in (("0", None), ("2", None), ("25", None), ("25", "1"), ("0", "0")):
    os.environ["MDX_NULL_WIRE_US"] = wire
    if pin is None: os.environ.pop("MDX_HALF_SHELL", None)
    else: os.environ["MDX_HALF_SHELL"] = pin
    s = systems.water_box(14, seed=6)
    try:
        with MdState(s, MdConfig(**CFG)) as md:
            md.comm_init_null(0, 8)
            d = md.comm_diag()
            print("attach:", d["half_shell"], d["wire_ns_measured"], md.stats()["repartitions"], flush=True)
            for k in range(3):
                md.step(0.0005, None, 1)
                st = md.stats()
                print(f"wire {wire} pin {pin}: step {k + 1}: rebuilds {st['rebuild_count']} local {st['local_rebuilds']} repartitions {st['repartitions']} T {md.energy()['temperature']:.0f}", flush=True)
    except Exception as e:
        print(f"wire {wire} pin {pin}: FAILED {str(e)[:300]}", flush=True)

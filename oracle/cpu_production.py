"""ctypes front-end of oracle/cpu_production.c — the fp32 "production mode" CPU baseline (half Verlet list reused
across steps, OpenMP) that bench.py times beside the GPU.  TEST / MEASUREMENT INFRASTRUCTURE ONLY: imported by
tests/ and bench.py's cpu_baseline leg, never by the product package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from molchanica_amd._abi import CConfig, CSystem, MdConfig, MdSystem

_HERE = os.path.dirname(os.path.abspath(__file__))
ENERGY_NAMES = ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14", "kinetic")
_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_libs = {}


def build(native: bool = False) -> str:
    """native=True compiles for the machine it runs on (-march=native): only ever done at run time on that machine,
    the portable build is what tests use."""
    target = "libcpuprod.so"
    if native:   # one build per CPU model: a -march=native object must never run on a machine it was not built on
        import hashlib
        try:
            flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags"))
        except Exception:
            flags = "unknown"
        target = "libcpuprod_native_%s.so" % hashlib.sha1(flags.encode()).hexdigest()[:10]
    path, src = os.path.join(_HERE, target), os.path.join(_HERE, "cpu_production.c")
    if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
        arch = "-march=native" if native else ""
        # (no -ffast-math: linking it into a shared object switches the whole process to flush-to-zero; the SIMD loops are
        # `omp simd` loops and vectorise without it)
        subprocess.check_call(f"gcc -O3 {arch} -fno-math-errno -fno-trapping-math -fPIC -shared -fopenmp -std=c11 "
                              f"{src} -o {path} -lm", shell=True)
    return path


def lib(native: bool = False):
    if native not in _libs:
        l = C.CDLL(build(native))
        l.cpu_prod_forces.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _fp, _fp, _dp]
        l.cpu_prod_run.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _fp, _fp, C.c_float, C.c_uint32, C.c_uint32, _dp]
        l.cpu_prod_max_threads.restype = C.c_int
        l.cpu_prod_last_pairs.restype = C.c_uint64
        _libs[native] = l
    return _libs[native]


def forces(sys: MdSystem, cfg: MdConfig, pos=None, native: bool = False):
    cs, cc = sys.to_c(), cfg.to_c()
    x = np.ascontiguousarray(sys.pos if pos is None else pos, dtype=np.float32).reshape(-1, 3).copy()
    f = np.zeros_like(x)
    en = np.zeros(len(ENERGY_NAMES))
    if lib(native).cpu_prod_forces(C.byref(cs), C.byref(cc), x.ctypes.data_as(_fp), f.ctypes.data_as(_fp), en.ctypes.data_as(_dp)) != 0:
        raise ValueError("cpu_production: unsupported system (needs a periodic box and cutoff / reaction-field Coulomb)")
    return f, dict(zip(ENERGY_NAMES, en))


def run(sys: MdSystem, cfg: MdConfig, dt: float, n_steps: int, energy_every: int = 0, pos=None, vel=None, native: bool = False):
    """-> (pos, vel, energies of the last evaluation, list builds)."""
    cs, cc = sys.to_c(), cfg.to_c()
    x = np.ascontiguousarray(sys.pos if pos is None else pos, dtype=np.float32).reshape(-1, 3).copy()
    v = np.ascontiguousarray(sys.vel if vel is None else vel, dtype=np.float32).reshape(-1, 3).copy()
    en = np.zeros(len(ENERGY_NAMES))
    rb = lib(native).cpu_prod_run(C.byref(cs), C.byref(cc), x.ctypes.data_as(_fp), v.ctypes.data_as(_fp), float(dt), int(n_steps),
                                  int(energy_every), en.ctypes.data_as(_dp))
    if rb < 0:
        raise ValueError("cpu_production: unsupported system")
    return x, v, dict(zip(ENERGY_NAMES, en)), rb

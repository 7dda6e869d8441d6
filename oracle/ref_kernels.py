"""ctypes front-end of oracle/_ref/libref_cuda.so: the REFERENCE's own pair kernels (/root/reference/src/cuda/cuda.cu +
util.cu) compiled for gfx950 by oracle/Makefile from where they lie.  TEST INFRASTRUCTURE ONLY.  The .so is prebuilt in
the build container (the sources do not travel to the GPU box); `available()` says whether it is there."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libref_cuda.so")
REFERENCE = "/root/reference"
_fp = C.POINTER(C.c_float)
_lib = None


def build() -> str | None:
    """Builds oracle/_ref/libref_cuda.so when the reference tree is present (this container only)."""
    if os.path.isdir(os.path.join(REFERENCE, "src", "cuda")):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return PATH if os.path.exists(PATH) else None


def available() -> bool:
    return os.path.exists(PATH)


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(PATH)
        l.ref_lj_force.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, _fp, _fp, _fp]
        l.ref_coulomb_force.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, _fp, C.c_size_t, _fp]
        l.ref_lj_V.argtypes = [_fp, C.c_size_t, _fp, C.c_size_t, C.c_float, C.c_float, _fp]
        l.ref_min_image.argtypes = [_fp, _fp, _fp]
        _lib = l
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def lj_force(tgt, src, sigma_ts, eps_ts):
    """`lj_force_kernel` (cuda.cu:73-102): force on every target from every source; sigma/eps tables [n_tgt, n_src]."""
    tgt, src, sg, ep = _f(tgt), _f(src), _f(sigma_ts), _f(eps_ts)
    out = np.zeros_like(tgt)
    rc = lib().ref_lj_force(tgt.ctypes.data_as(_fp), len(tgt), src.ctypes.data_as(_fp), len(src), sg.ctypes.data_as(_fp), ep.ctypes.data_as(_fp),
                            out.ctypes.data_as(_fp))
    assert rc == 0, rc
    return out


def coulomb_force(tgt, src, charges):
    """`coulomb_force_kernel` (cuda.cu:10-37): source i carries charges[i], target j carries charges[j] (one array)."""
    tgt, src, q = _f(tgt), _f(src), _f(charges)
    out = np.zeros_like(tgt)
    rc = lib().ref_coulomb_force(tgt.ctypes.data_as(_fp), len(tgt), src.ctypes.data_as(_fp), len(src), q.ctypes.data_as(_fp), len(q),
                                 out.ctypes.data_as(_fp))
    assert rc == 0, rc
    return out


def lj_V(src, tgt, sigma, eps):
    """`lj_V_kernel` (cuda.cu:40-70): per target, the LJ energy with all sources (one sigma, one eps)."""
    src, tgt = _f(src), _f(tgt)
    out = np.zeros(len(tgt), np.float32)
    rc = lib().ref_lj_V(src.ctypes.data_as(_fp), len(src), tgt.ctypes.data_as(_fp), len(tgt), float(sigma), float(eps), out.ctypes.data_as(_fp))
    assert rc == 0, rc
    return out


def min_image(ext, dv):
    """`min_image` (util.cu:65-71)."""
    e, d, o = _f(ext), _f(dv), np.zeros(3, np.float32)
    rc = lib().ref_min_image(e.ctypes.data_as(_fp), d.ctypes.data_as(_fp), o.ctypes.data_as(_fp))
    assert rc == 0, rc
    return o

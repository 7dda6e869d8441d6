"""ctypes front-end of oracle/liborc.so (CPU restatement; TEST INFRASTRUCTURE ONLY — see the
header of mdx_oracle.c.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module; the product package molchanica_amd never does)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from molchanica_amd._abi import CConfig, CSystem, MdConfig, MdSystem

_HERE = os.path.dirname(os.path.abspath(__file__))
ENERGY_NAMES = ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14", "kinetic", "virial", "cross", "dudl", "gross_lj", "gross_coulomb")
BAR_PER_KCAL_MOL_A3 = 69476.95

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def build(extra: str = "", target: str = "liborc.so") -> str:
    path = os.path.join(_HERE, target)
    src = os.path.join(_HERE, "mdx_oracle.c")
    if (not os.path.exists(path)) or os.path.getmtime(path) < os.path.getmtime(src):
        if target == "liborc.so" and not extra:
            subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
        else:
            subprocess.check_call(
                f"gcc -O3 {extra} -fPIC -shared -fopenmp -ffp-contract=off -std=c11 "
                f"{src} -o {path} -lm", shell=True)
    return path


_lib = None


def lib(path: str | None = None):
    global _lib
    if _lib is None or path is not None:
        l = C.CDLL(path or build())
        l.orc_forces.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _dp, _dp, _dp, _dp, C.c_int]
        l.orc_step.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _dp, _dp, C.c_double,
                               C.c_uint32, _dp, _dp, C.c_int]
        l.orc_neighbor_list.argtypes = [C.POINTER(CSystem), _fp, C.c_float, _u32p, _u32p, C.c_int]
        l.orc_cutoff_slack.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _fp, C.c_double, _dp]
        l.orc_kinetic.argtypes = [C.POINTER(CSystem), _dp]
        l.orc_kinetic.restype = C.c_double
        l.orc_wrap_f32.argtypes = [C.POINTER(CSystem), _fp, C.c_uint32]
        l.orc_min_image_f32.argtypes = [C.POINTER(CSystem), _fp, _fp]
        l.orc_min_image_f32.restype = None
        l.orc_r2_canonical.argtypes = [C.POINTER(CSystem), _fp, _fp]
        l.orc_r2_canonical.restype = C.c_float
        l.orc_constrain_positions.argtypes = [C.POINTER(CSystem), _dp, _dp, _dp, C.c_double, C.c_double]
        l.orc_constrain_velocities.argtypes = [C.POINTER(CSystem), _dp, _dp, C.c_double]
        l.orc_vsite_construct.argtypes = [C.POINTER(CSystem), _dp]
        l.orc_vsite_construct.restype = None
        l.orc_dof.argtypes = [C.POINTER(CSystem)]
        l.orc_dof.restype = C.c_double
        l.orc_init_velocities.argtypes = [C.POINTER(CSystem), C.c_double, C.c_int, C.c_uint64, _dp]
        l.orc_init_velocities.restype = None
        l.orc_thermostat_lambda.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double,
                                            C.POINTER(C.c_uint64)]
        l.orc_thermostat_lambda.restype = C.c_double
        l.orc_step_thermo.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _dp, _dp, C.c_double, C.c_uint32,
                                      C.c_int, C.c_double, C.c_double, C.c_uint32, C.c_uint64, C.c_int, _dp, C.c_int]
        l.orc_between_mols.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _dp, C.c_void_p, C.c_uint32, _dp, _dp, C.c_int]
        l.orc_minimize.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), _dp, C.c_uint32, _dp, C.c_double, _dp, C.c_int]
        if path is not None:
            return l
        _lib = l
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def forces(sys: MdSystem, cfg: MdConfig, pos=None, ext=None, use_cells=False, _lib_override=None):
    """-> (forces [N,3] f64, energies dict).  pos: optional fp64 [N,3] override."""
    l = _lib_override or lib()
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    x = None if pos is None else np.ascontiguousarray(pos, dtype=np.float64).reshape(n, 3)
    e = None if ext is None else np.ascontiguousarray(ext, dtype=np.float64).reshape(n, 3)
    f = np.zeros((n, 3), dtype=np.float64)
    en = np.zeros(len(ENERGY_NAMES), dtype=np.float64)
    rc = l.orc_forces(C.byref(cs), C.byref(cc), _d(x), _d(e), _d(f), _d(en), int(use_cells))
    assert rc == 0
    return f, _energies(en)


def between_mols(sys: MdSystem, cfg: MdConfig, group, n_groups: int, pos=None, use_cells=False):
    """energy_potential_between_mols (/root/reference src/properties/crystal.rs:347-370, 533) -> (matrix [G,G] f64, gross [G,G])."""
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    x = None if pos is None else np.ascontiguousarray(pos, dtype=np.float64).reshape(n, 3)
    g = np.ascontiguousarray(group, dtype=np.uint8).reshape(n)
    assert int(g.max()) < n_groups
    m = np.zeros((n_groups, n_groups)); gr = np.zeros((n_groups, n_groups))
    rc = lib().orc_between_mols(C.byref(cs), C.byref(cc), _d(x), g.ctypes.data, int(n_groups), _d(m), _d(gr), int(use_cells))
    assert rc == 0
    return m, gr


def _energies(en):
    assert lib().orc_num_energies() == len(ENERGY_NAMES)
    d = dict(zip(ENERGY_NAMES, (float(v) for v in en)))
    d["potential_bonded"] = d["bond"] + d["angle"] + d["dihedral"]
    d["potential_nonbonded"] = d["lj"] + d["coulomb"] + d["lj14"] + d["coulomb14"]
    d["potential"] = d["potential_bonded"] + d["potential_nonbonded"]
    return d


def step(sys: MdSystem, cfg: MdConfig, dt: float, n_steps: int, pos=None, vel=None, ext=None,
         use_cells=False):
    """Velocity Verlet in fp64.  -> (pos, vel, energies-of-final-state)."""
    l = lib()
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    x = np.array(sys.pos if pos is None else pos, dtype=np.float64).reshape(n, 3).copy()
    if vel is None:
        vel = sys.vel if sys.vel is not None else np.zeros((n, 3))
    v = np.array(vel, dtype=np.float64).reshape(n, 3).copy()
    e = None if ext is None else np.ascontiguousarray(ext, dtype=np.float64).reshape(n, 3)
    en = np.zeros(len(ENERGY_NAMES), dtype=np.float64)
    rc = l.orc_step(C.byref(cs), C.byref(cc), _d(x), _d(v), float(dt), int(n_steps), _d(e), _d(en),
                    int(use_cells))
    assert rc == 0
    return x, v, _energies(en)


def neighbor_list(sys: MdSystem, rlist: float, pos=None, use_cells=False):
    """-> (offsets [N+1] u32, idx u32): canonical-fp32 Verlet list, rows ascending."""
    l = lib()
    cs = sys.to_c()
    n = sys.n_atoms
    p = np.ascontiguousarray(sys.pos if pos is None else pos, dtype=np.float32).reshape(n, 3)
    off = np.zeros(n + 1, dtype=np.uint32)
    l.orc_neighbor_list(C.byref(cs), p.ctypes.data_as(_fp), float(rlist), off.ctypes.data_as(_u32p),
                        None, int(use_cells))
    idx = np.zeros(max(int(off[-1]), 1), dtype=np.uint32)
    l.orc_neighbor_list(C.byref(cs), p.ctypes.data_as(_fp), float(rlist), off.ctypes.data_as(_u32p),
                        idx.ctypes.data_as(_u32p), int(use_cells))
    return off, idx[: int(off[-1])]


def cutoff_slack(sys: MdSystem, cfg: MdConfig, pos=None, rel: float = 1e-5):
    l = lib()
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    p = np.ascontiguousarray(sys.pos if pos is None else pos, dtype=np.float32).reshape(n, 3)
    out = np.zeros(n, dtype=np.float64)
    l.orc_cutoff_slack(C.byref(cs), C.byref(cc), p.ctypes.data_as(_fp), float(rel), _d(out))
    return out


def min_image_f32(sys: MdSystem, d):
    """The oracle's canonical fp32 minimum image of a difference vector (the arithmetic inside r2_canonical)."""
    l = lib()
    cs = sys.to_c()
    di = np.ascontiguousarray(d, dtype=np.float32).reshape(3)
    out = np.zeros(3, np.float32)
    l.orc_min_image_f32(C.byref(cs), di.ctypes.data_as(_fp), out.ctypes.data_as(_fp))
    return out


def r2_canonical(sys: MdSystem, pi, pj) -> float:
    l = lib()
    cs = sys.to_c()
    a = np.ascontiguousarray(pi, dtype=np.float32).reshape(3); b = np.ascontiguousarray(pj, dtype=np.float32).reshape(3)
    return float(l.orc_r2_canonical(C.byref(cs), a.ctypes.data_as(_fp), b.ctypes.data_as(_fp)))


def wrap(sys: MdSystem, pos):
    l = lib()
    cs = sys.to_c()
    p = np.array(pos, dtype=np.float32).reshape(-1, 3).copy()
    l.orc_wrap_f32(C.byref(cs), p.ctypes.data_as(_fp), p.shape[0])
    return p


def kinetic(sys: MdSystem, vel):
    l = lib()
    cs = sys.to_c()
    v = np.ascontiguousarray(vel, dtype=np.float64).reshape(-1, 3)
    return float(l.orc_kinetic(C.byref(cs), _d(v)))


def dof(sys: MdSystem) -> float:
    cs = sys.to_c()
    return float(lib().orc_dof(C.byref(cs)))


def init_velocities(sys: MdSystem, temperature: float, zero_com: bool, seed: int):
    cs = sys.to_c()
    v = np.zeros((sys.n_atoms, 3), dtype=np.float64)
    lib().orc_init_velocities(C.byref(cs), float(temperature), int(zero_com), int(seed), _d(v))
    return v


def step_thermo(sys: MdSystem, cfg: MdConfig, dt, n_steps, kind, temp_target, tau, every, seed, zero_com=False,
                pos=None, vel=None, use_cells=False):
    """-> (pos, vel, temperatures after each coupling)."""
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    x = np.array(sys.pos if pos is None else pos, dtype=np.float64).reshape(n, 3).copy()
    v = np.array(sys.vel if vel is None else vel, dtype=np.float64).reshape(n, 3).copy()
    temps = np.zeros(max(n_steps // every, 1), dtype=np.float64)
    lib().orc_step_thermo(C.byref(cs), C.byref(cc), _d(x), _d(v), float(dt), int(n_steps), int(kind),
                          float(temp_target), float(tau), int(every), int(seed), int(zero_com), _d(temps),
                          int(use_cells))
    return x, v, temps[: n_steps // every]


def minimize(sys: MdSystem, cfg: MdConfig, max_iters, f_tol=0.0, pos=None, ext=None, use_cells=False):
    """-> (pos, energies, trial evaluations)."""
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    x = np.array(sys.pos if pos is None else pos, dtype=np.float64).reshape(n, 3).copy()
    e = None if ext is None else np.ascontiguousarray(ext, dtype=np.float64).reshape(n, 3)
    en = np.zeros(len(ENERGY_NAMES), dtype=np.float64)
    it = lib().orc_minimize(C.byref(cs), C.byref(cc), _d(x), int(max_iters), _d(e), float(f_tol), _d(en),
                            int(use_cells))
    return x, _energies(en), int(it)


def constrain_positions(sys: MdSystem, x_new, x_old, vel=None, dt=0.0, tol=1e-12):
    cs = sys.to_c()
    x = np.array(x_new, dtype=np.float64).reshape(-1, 3).copy()
    xo = np.ascontiguousarray(x_old, dtype=np.float64).reshape(-1, 3)
    v = None if vel is None else np.array(vel, dtype=np.float64).reshape(-1, 3).copy()
    it = lib().orc_constrain_positions(C.byref(cs), _d(x), _d(xo), _d(v), float(dt), float(tol))
    return x, v, int(it)


def constrain_velocities(sys: MdSystem, x, vel, tol=1e-12):
    cs = sys.to_c()
    xx = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, 3)
    v = np.array(vel, dtype=np.float64).reshape(-1, 3).copy()
    lib().orc_constrain_velocities(C.byref(cs), _d(xx), _d(v), float(tol))
    return v


def vsite_construct(sys: MdSystem, x):
    cs = sys.to_c()
    xx = np.array(x, dtype=np.float64).reshape(-1, 3).copy()
    lib().orc_vsite_construct(C.byref(cs), _d(xx))
    return xx


def last_constraint_virial() -> float:
    """sum r . G of the SHAKE forces of the most recent position stage (kcal/mol)."""
    l = lib()
    l.orc_last_constraint_virial.restype = C.c_double
    return float(l.orc_last_constraint_virial())


def pressure(sys: MdSystem, energies: dict, kinetic_energy: float, w_constraints: float = 0.0) -> float:
    """bar: (2 KE + W + W_constraints) / (3 V) x 69476.95, V from sys.box."""
    vol = float(np.prod(np.asarray(sys.box_hi, np.float64) - np.asarray(sys.box_lo, np.float64)))
    return (2.0 * kinetic_energy + energies["virial"] + w_constraints) / (3.0 * vol) * BAR_PER_KCAL_MOL_A3


def step_npt(sys: MdSystem, cfg: MdConfig, dt, n_steps, pos=None, vel=None, thermostat=(0, 300.0, 1.0, 10, 0),
             barostat=(1, 1.0, 5.0, 4.5e-5, 25), use_cells=False):
    """thermostat = (kind, T, tau, every, seed); barostat = (kind, P0 bar, tau, compressibility, every).
    -> (pos, vel, box_hi, pressures, volumes) with one pressure/volume per barostat application."""
    l = lib()
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    x = np.array(sys.pos if pos is None else pos, dtype=np.float64).reshape(n, 3).copy()
    v = np.array(sys.vel if vel is None else vel, dtype=np.float64).reshape(n, 3).copy()
    hi = np.array(sys.box_hi, dtype=np.float32).copy()
    nb = max(1, n_steps // max(1, barostat[4]) + 1)
    ps = np.zeros(nb); vs = np.zeros(nb)
    k = l.orc_step_npt(C.byref(cs), C.byref(cc), _d(x), _d(v), C.c_double(dt), C.c_uint32(n_steps),
                       C.c_int(thermostat[0]), C.c_double(thermostat[1]), C.c_double(thermostat[2]), C.c_uint32(thermostat[3]),
                       C.c_uint64(thermostat[4]),
                       C.c_int(barostat[0]), C.c_double(barostat[1]), C.c_double(barostat[2]), C.c_double(barostat[3]),
                       C.c_uint32(barostat[4]), hi.ctypes.data_as(C.POINTER(C.c_float)), _d(ps), _d(vs), C.c_int(int(use_cells)))
    return x, v, hi, ps[:k], vs[:k]


def step_integrator(sys: MdSystem, cfg: MdConfig, dt, n_steps, kind, gamma=1.0, temperature=300.0, seed=0, step0=0,
                    pos=None, vel=None, use_cells=False):
    """kind 1 leapfrog, 2 Langevin middle.  -> (pos, half-step vel, energies of the final state)."""
    l = lib()
    cs, cc = sys.to_c(), cfg.to_c()
    n = sys.n_atoms
    x = np.array(sys.pos if pos is None else pos, dtype=np.float64).reshape(n, 3).copy()
    v = np.array(sys.vel if vel is None else vel, dtype=np.float64).reshape(n, 3).copy()
    en = np.zeros(len(ENERGY_NAMES), dtype=np.float64)
    rc = l.orc_step_integrator(C.byref(cs), C.byref(cc), _d(x), _d(v), C.c_double(dt), C.c_uint32(n_steps), C.c_int(kind),
                               C.c_double(gamma), C.c_double(temperature), C.c_uint64(seed), C.c_uint64(step0), _d(en),
                               C.c_int(int(use_cells)))
    assert rc == 0
    return x, v, _energies(en)


def langevin_normals(seed, step, atom):
    g = np.zeros(3)
    lib().orc_langevin_normals(C.c_uint64(seed), C.c_uint64(step), C.c_uint32(atom), _d(g))
    return g


def set_alchemical(lo: int, hi: int, lam: float):
    """Couple atoms [lo, hi) to the rest with factor (1 - lam) on their mutual non-bonded pairs; lam < 0 = off.
    Energies then carry "cross" (unscaled U_cross): dU/dlambda = -cross."""
    lib().orc_set_alchemical(C.c_uint32(int(lo)), C.c_uint32(int(hi)), C.c_double(float(lam)))


def set_softcore(alpha: float, sigma_min: float = 3.0):
    """Soft-core coupling of the alchemical window: cross pairs interact at r_sc = (alpha sigma^6 lambda + r^6)^(1/6);
    alpha = 0 (the oracle's default) is the linear coupling.  Energies carry "dudl" = dU/dlambda either way."""
    lib().orc_set_softcore(C.c_double(float(alpha)), C.c_double(float(sigma_min)))


def shrink_cell_towards(box_lo, box_hi, target_lo, target_hi, shrink_per_step, pos):
    """`md.shrink_cell_towards` as the reference states the rule for its GROMACS backend
    (/root/reference src/properties/sol_shrinking_box.rs:765-774 `shrink_cell_by_amount`: every edge shrinks by the amount
    but not below the limit's, the cell keeps its centre) with the coordinates following affinely (its `deform`,
    :1263-1275).  fp32 cell arithmetic like the engine's SimBox; -> (lo, hi, pos, shrank)."""
    lo = np.asarray(box_lo, np.float32); hi = np.asarray(box_hi, np.float32)
    e = hi - lo
    te = np.asarray(target_hi, np.float32) - np.asarray(target_lo, np.float32)
    ne = np.maximum(e - np.float32(shrink_per_step), te).astype(np.float32)
    c = (np.float32(0.5) * (lo + hi)).astype(np.float32)
    nlo = (c - np.float32(0.5) * ne).astype(np.float32); nhi = (c + np.float32(0.5) * ne).astype(np.float32)
    mu = (ne / e).astype(np.float32)
    x = np.asarray(pos, np.float64)
    return nlo, nhi, c.astype(np.float64) + mu.astype(np.float64) * (x - c.astype(np.float64)), bool((ne != e).any())

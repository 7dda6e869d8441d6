"""ctypes mirror of include/mdx.h (struct layouts and constants).

Kept free of any library loading so that CPU-only tooling and tests can use the same struct
definitions without a GPU build being present.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

MDX_OK, MDX_EPARAM, MDX_EDEVICE, MDX_ENAN, MDX_EOOM = 0, -1, -2, -3, -4

OVR_BONDED_DISABLED = 0x1
OVR_COULOMB_DISABLED = 0x2
OVR_LJ_DISABLED = 0x4
OVR_LONG_RANGE_RECIP_DISABLED = 0x8

ATOM_STATIC, ATOM_BONDED_ONLY, ATOM_GHOST = 0x1, 0x2, 0x4

COULOMB_SHIFTED, COULOMB_REACTION, COULOMB_EWALD = 0, 1, 2
COMBINE_LORENTZ_BERTHELOT, COMBINE_GEOMETRIC = 0, 1

POS, VEL, FORCE = 0, 1, 2

_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


class CSystem(C.Structure):
    _fields_ = [
        ("n_atoms", C.c_uint32),
        ("pos", _fp), ("vel", _fp), ("mass", _fp), ("charge", _fp),
        ("lj_type", _u32p), ("n_lj_types", C.c_uint32),
        ("lj_sigma", _fp), ("lj_eps", _fp), ("flags", _u8p),
        ("n_bonds", C.c_uint32), ("bond_idx", _u32p), ("bond_k", _fp), ("bond_r0", _fp),
        ("n_angles", C.c_uint32), ("angle_idx", _u32p), ("angle_k", _fp), ("angle_theta0", _fp),
        ("n_dihedrals", C.c_uint32), ("dihedral_idx", _u32p), ("dihedral_v", _fp),
        ("dihedral_phase", _fp), ("dihedral_n", _i32p),
        ("excl_offsets", _u32p), ("excl_idx", _u32p),
        ("n_pairs14", C.c_uint32), ("pairs14_idx", _u32p),
        ("n_mols", C.c_uint32), ("mol_start", _u32p),
        ("periodic", C.c_int32), ("box_lo", C.c_float * 3), ("box_hi", C.c_float * 3),
        ("n_constraints", C.c_uint32), ("constraint_idx", _u32p), ("constraint_len", _fp),
        ("n_vsites", C.c_uint32), ("vsite_idx", _u32p), ("vsite_w", _fp),
    ]


class CConfig(C.Structure):
    _fields_ = [
        ("lj_cutoff", C.c_float), ("coulomb_cutoff", C.c_float), ("skin", C.c_float),
        ("coulomb_k", C.c_float), ("scale14_lj", C.c_float), ("scale14_coulomb", C.c_float),
        ("coulomb_mode", C.c_int32), ("ewald_alpha", C.c_float), ("combining_rule", C.c_int32),
        ("overrides", C.c_uint32), ("softening_sq", C.c_float), ("chunk_steps", C.c_uint32),
        ("nb_variant", C.c_uint32), ("constraint_tol", C.c_float), ("constraint_max_iter", C.c_uint32),
        ("pme_grid", C.c_uint32 * 3), ("pme_order", C.c_uint32), ("inner_skin", C.c_float),
    ]


class CEnergies(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "kinetic", "potential", "potential_nonbonded", "potential_bonded",
        "lj", "coulomb", "lj14", "coulomb14", "bond", "angle", "dihedral",
        "temperature", "volume", "density", "virial", "max_force", "coulomb_recip", "dh_dlambda", "coupled_interaction", "pressure")]

    def as_dict(self) -> dict:
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


class CHBond(C.Structure):
    """mdx_hbond: (HBondAtomType, index) references of donor / acceptor / hydrogen + strength."""
    _fields_ = [("donor", C.c_uint32), ("acceptor", C.c_uint32), ("hydrogen", C.c_uint32),
                ("donor_type", C.c_uint8), ("acceptor_type", C.c_uint8), ("hydrogen_type", C.c_uint8), ("pad", C.c_uint8),
                ("strength", C.c_float)]


DIAG_PHASES = ("halo_pack", "halo_wire", "halo_unpack", "force_pack", "force_wire", "force_add", "pair", "pair_boundary",
               "bonded", "integrate")


class CCommDiag(C.Structure):
    """mdx_comm_diag (include/mdx.h): where a decomposed step spends its time and what it moves."""
    _fields_ = [
        ("transport", C.c_char * 64), ("rank", C.c_int32), ("world", C.c_int32), ("grid", C.c_int32 * 3),
        ("rccl_version", C.c_int32), ("rccl_comm_count", C.c_int32), ("half_shell", C.c_int32),
        ("overlap_split", C.c_int32), ("comm_stream_separate", C.c_int32), ("wire_ns_measured", C.c_int32),
        ("n_owned", C.c_uint32), ("n_ghost", C.c_uint32), ("n_tiles", C.c_uint32), ("n_interior_tiles", C.c_uint32),
        ("halo_rows_out", C.c_uint32), ("halo_rows_in", C.c_uint32), ("halo_bytes_per_step", C.c_uint64),
        ("repartitions", C.c_uint64), ("local_rebuilds", C.c_uint64), ("repartition_ms_sum", C.c_double),
        ("phase_ms", C.c_double * len(DIAG_PHASES)), ("phase_n", C.c_uint64 * len(DIAG_PHASES)),
    ]

    def as_dict(self) -> dict:
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("transport", "grid", "phase_ms", "phase_n")}
        d["transport"] = self.transport.decode()
        d["grid"] = tuple(self.grid)
        d["phase_ms_sum"] = {k: float(self.phase_ms[i]) for i, k in enumerate(DIAG_PHASES)}
        d["phase_brackets"] = {k: int(self.phase_n[i]) for i, k in enumerate(DIAG_PHASES)}
        return d


class CStats(C.Structure):
    _fields_ = [
        ("step_count", C.c_uint64), ("rebuild_count", C.c_uint64),
        ("n_atoms", C.c_uint32), ("n_slots", C.c_uint32), ("n_tiles", C.c_uint32),
        ("n_clusters", C.c_uint32),
        ("n_list_entries", C.c_uint64), ("n_masked_entries", C.c_uint64),
        ("n_cluster_pairs", C.c_uint64),
        ("nb_ms_sum", C.c_double), ("nb_launches", C.c_uint64),
        ("bonded_ms_sum", C.c_double), ("bonded_launches", C.c_uint64),
        ("integ_ms_sum", C.c_double), ("integ_launches", C.c_uint64),
        ("rebuild_ms_sum", C.c_double), ("wall_ms_sum", C.c_double),
        ("n_inner_cluster_pairs", C.c_uint64), ("prune_passes", C.c_uint64),
        ("n_owned", C.c_uint32), ("n_ghost", C.c_uint32), ("repartitions", C.c_uint64),
        ("local_rebuilds", C.c_uint64), ("repartition_ms_sum", C.c_double),
        ("fused_ms_sum", C.c_double), ("fused_launches", C.c_uint64),
        ("energy_evaluations", C.c_uint64), ("energies_from_step_loop", C.c_uint64),
        ("rebuild_fallbacks", C.c_uint64),
    ]

    def as_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_}


@dataclass
class MdConfig:
    """The force/integrate subset of the reference's `MdConfig`
    (src/ui/panels/md.rs:252-261, 291-305; overrides src/md/mod.rs:671-686)."""
    lj_cutoff: float = 10.0
    coulomb_cutoff: float = 10.0
    skin: float = 2.0
    coulomb_k: float = 332.0637
    scale14_lj: float = 0.5
    scale14_coulomb: float = 1.0 / 1.2
    coulomb_mode: int = COULOMB_SHIFTED
    ewald_alpha: float = 0.0
    combining_rule: int = COMBINE_LORENTZ_BERTHELOT
    overrides: int = OVR_LONG_RANGE_RECIP_DISABLED
    softening_sq: float = 0.0
    chunk_steps: int = 16
    nb_variant: int = 0
    constraint_tol: float = 1e-5
    constraint_max_iter: int = 64
    pme_grid: tuple = (0, 0, 0)
    pme_order: int = 4
    inner_skin: float = 0.0     # dual pair list buffer; 0 = library default (0.5 A), < 0 = off

    def to_c(self) -> CConfig:
        c = CConfig()
        for k in ("lj_cutoff", "coulomb_cutoff", "skin", "coulomb_k", "scale14_lj",
                  "scale14_coulomb", "coulomb_mode", "ewald_alpha", "combining_rule",
                  "overrides", "softening_sq", "chunk_steps", "nb_variant", "constraint_tol",
                  "constraint_max_iter"):
            setattr(c, k, getattr(self, k))
        c.pme_grid = (C.c_uint32 * 3)(*[int(v) for v in self.pme_grid])
        c.pme_order = int(self.pme_order)
        c.inner_skin = float(self.inner_skin)
        return c


def _arr(a, dtype, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
    if shape is not None:
        a = a.reshape(shape)
    return a


class SimBoxInit:
    """`SimBoxInit::{Pad(f32), Fixed((lo, hi))}` and `SimBoxInit::new_cube(side)` (/root/reference src/md/mod.rs:656-659,
    src/ui/panels/md.rs:582-584, src/properties/water_sol.rs:190, crystal.rs:321, logp.rs:119).  Pad: the extent of the atoms plus
    `pad` on every side - the rule the reference restates itself at src/gromacs/mod.rs:540-576 ("a copy+paste of SimBox::from_atoms
    in dynamics").  new_cube: a cube of the given edge about `centre` (where the crate centres it is not visible in the tree)."""

    def __init__(self, kind: str, pad: float = 12.0, lo=None, hi=None):
        assert kind in ("pad", "fixed")
        self.kind, self.pad = kind, float(pad)
        self.lo = None if lo is None else np.asarray(lo, np.float32).reshape(3)
        self.hi = None if hi is None else np.asarray(hi, np.float32).reshape(3)

    @classmethod
    def Pad(cls, pad: float) -> "SimBoxInit":
        return cls("pad", pad=pad)

    @classmethod
    def Fixed(cls, lo, hi) -> "SimBoxInit":
        return cls("fixed", lo=lo, hi=hi)

    @classmethod
    def new_cube(cls, side: float, centre=(0.0, 0.0, 0.0)) -> "SimBoxInit":
        c = np.asarray(centre, np.float32)
        return cls.Fixed(c - np.float32(0.5 * side), c + np.float32(0.5 * side))

    def resolve(self, pos):
        """`SimBox::from_atoms` -> (bounds_low, bounds_high), float32[3] each."""
        if self.kind == "fixed":
            return self.lo.copy(), self.hi.copy()
        p = np.asarray(pos, np.float32).reshape(-1, 3)
        if p.shape[0] == 0:
            raise ValueError("SimBoxInit.Pad needs at least one atom")        # (`sim_box_nm` returns None, src/gromacs/mod.rs:571-573)
        return p.min(0) - np.float32(self.pad), p.max(0) + np.float32(self.pad)


@dataclass
class MdSystem:
    """Flat SoA system: the shape `setup_mols_dyn` hands to `MdState::new`
    (src/md/mod.rs:1076-1157: atoms, atom_posits, atom_init_velocities, bonds, static_,
    bonded_only), after parameterisation."""
    pos: np.ndarray                       # [N,3] f32
    mass: np.ndarray                      # [N]
    charge: np.ndarray                    # [N]
    lj_type: np.ndarray                   # [N] u32
    lj_sigma: np.ndarray                  # [T]
    lj_eps: np.ndarray                    # [T]
    vel: np.ndarray | None = None         # [N,3]
    flags: np.ndarray | None = None       # [N] u8
    bond_idx: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))
    bond_k: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    bond_r0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    angle_idx: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.uint32))
    angle_k: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    angle_theta0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    dihedral_idx: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.uint32))
    dihedral_v: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    dihedral_phase: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    dihedral_n: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    excl_offsets: np.ndarray | None = None  # [N+1]
    excl_idx: np.ndarray | None = None
    pairs14_idx: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))
    mol_start: np.ndarray | None = None
    constraint_idx: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.uint32))
    constraint_len: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    vsite_idx: np.ndarray = field(default_factory=lambda: np.zeros((0, 4), np.uint32))
    vsite_w: np.ndarray = field(default_factory=lambda: np.zeros((0, 2), np.float32))
    periodic: bool = False
    box_lo: tuple = (0.0, 0.0, 0.0)
    box_hi: tuple = (0.0, 0.0, 0.0)
    name: str = ""

    @property
    def n_atoms(self) -> int:
        return int(np.asarray(self.pos).reshape(-1, 3).shape[0])

    def apply_sim_box(self, init: "SimBoxInit") -> "MdSystem":
        """`cfg.sim_box` as `MdState::new` applies it (src/md/mod.rs:656-659): the cell of these atoms, periodic."""
        lo, hi = init.resolve(self.pos)
        self.box_lo, self.box_hi, self.periodic = tuple(float(v) for v in lo), tuple(float(v) for v in hi), True
        return self

    def normalise(self) -> "MdSystem":
        n = self.n_atoms
        self.pos = _arr(self.pos, np.float32, (n, 3))
        self.mass = _arr(self.mass, np.float32, (n,))
        self.charge = _arr(self.charge, np.float32, (n,))
        self.lj_type = _arr(self.lj_type, np.uint32, (n,))
        self.lj_sigma = _arr(self.lj_sigma, np.float32)
        self.lj_eps = _arr(self.lj_eps, np.float32)
        if self.vel is not None:
            self.vel = _arr(self.vel, np.float32, (n, 3))
        if self.flags is not None:
            self.flags = _arr(self.flags, np.uint8, (n,))
        self.bond_idx = _arr(self.bond_idx, np.uint32, (-1, 2))
        self.bond_k = _arr(self.bond_k, np.float32)
        self.bond_r0 = _arr(self.bond_r0, np.float32)
        self.angle_idx = _arr(self.angle_idx, np.uint32, (-1, 3))
        self.angle_k = _arr(self.angle_k, np.float32)
        self.angle_theta0 = _arr(self.angle_theta0, np.float32)
        self.dihedral_idx = _arr(self.dihedral_idx, np.uint32, (-1, 4))
        self.dihedral_v = _arr(self.dihedral_v, np.float32)
        self.dihedral_phase = _arr(self.dihedral_phase, np.float32)
        self.dihedral_n = _arr(self.dihedral_n, np.int32)
        if self.excl_offsets is None:
            self.excl_offsets = np.zeros(n + 1, np.uint32)
            self.excl_idx = np.zeros(0, np.uint32)
        self.excl_offsets = _arr(self.excl_offsets, np.uint32, (n + 1,))
        self.excl_idx = _arr(self.excl_idx, np.uint32)
        self.pairs14_idx = _arr(self.pairs14_idx, np.uint32, (-1, 2))
        if self.mol_start is not None:
            self.mol_start = _arr(self.mol_start, np.uint32)
        self.constraint_idx = _arr(self.constraint_idx, np.uint32, (-1, 2))
        self.constraint_len = _arr(self.constraint_len, np.float32)
        self.vsite_idx = _arr(self.vsite_idx, np.uint32, (-1, 4))
        self.vsite_w = _arr(self.vsite_w, np.float32, (-1, 2))
        return self

    def to_c(self) -> CSystem:
        """Returns a CSystem whose pointers alias this object's arrays (keep `self` alive)."""
        self.normalise()
        s = CSystem()

        def p(a, t):
            if a is None or a.size == 0:
                return C.cast(None, t)
            return a.ctypes.data_as(t)

        s.n_atoms = self.n_atoms
        s.pos, s.vel = p(self.pos, _fp), p(self.vel, _fp)
        s.mass, s.charge = p(self.mass, _fp), p(self.charge, _fp)
        s.lj_type, s.n_lj_types = p(self.lj_type, _u32p), int(self.lj_sigma.size)
        s.lj_sigma, s.lj_eps = p(self.lj_sigma, _fp), p(self.lj_eps, _fp)
        s.flags = p(self.flags, _u8p)
        s.n_bonds = int(self.bond_idx.shape[0])
        s.bond_idx, s.bond_k, s.bond_r0 = p(self.bond_idx, _u32p), p(self.bond_k, _fp), p(self.bond_r0, _fp)
        s.n_angles = int(self.angle_idx.shape[0])
        s.angle_idx, s.angle_k, s.angle_theta0 = (p(self.angle_idx, _u32p), p(self.angle_k, _fp),
                                                  p(self.angle_theta0, _fp))
        s.n_dihedrals = int(self.dihedral_idx.shape[0])
        s.dihedral_idx, s.dihedral_v = p(self.dihedral_idx, _u32p), p(self.dihedral_v, _fp)
        s.dihedral_phase, s.dihedral_n = p(self.dihedral_phase, _fp), p(self.dihedral_n, _i32p)
        s.excl_offsets = self.excl_offsets.ctypes.data_as(_u32p)
        s.excl_idx = p(self.excl_idx, _u32p)
        s.n_pairs14, s.pairs14_idx = int(self.pairs14_idx.shape[0]), p(self.pairs14_idx, _u32p)
        s.n_mols = 0 if self.mol_start is None else int(self.mol_start.size)
        s.mol_start = p(self.mol_start, _u32p)
        s.periodic = 1 if self.periodic else 0
        s.box_lo = (C.c_float * 3)(*[float(v) for v in self.box_lo])
        s.box_hi = (C.c_float * 3)(*[float(v) for v in self.box_hi])
        s.n_constraints = int(self.constraint_idx.shape[0])
        s.constraint_idx, s.constraint_len = p(self.constraint_idx, _u32p), p(self.constraint_len, _fp)
        s.n_vsites = int(self.vsite_idx.shape[0])
        s.vsite_idx, s.vsite_w = p(self.vsite_idx, _u32p), p(self.vsite_w, _fp)
        return s

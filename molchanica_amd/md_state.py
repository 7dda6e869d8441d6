"""Host-side mirror of the `dynamics` crate surface Molchanica calls, over the C ABI (libmdx.so).

Names, argument meaning and error behaviour follow the reference call sites
(/root/reference, `dynamics = "0.2.2"` is external):
  MdState::new(dev, &cfg, &mols, param_set) -> Result<(MdState, ..), ParamError>   src/md/mod.rs:689
  md.step(&dev, dt, Option<Vec<Vec3F32>>)                                           src/md/mod.rs:748
  run_dynamics_blocking(md, dev, dt, n_steps)                                       src/md/mod.rs:696-724
  compute_energy_snapshot(dev, &mols, param_set) -> Result<Snapshot, ParamError>    src/md/mod.rs:1036
  md.atoms[i].posit / .force, md.step_count, md.cell, md.rebuild_spatial_caches()   src/mol_alignment.rs:349-352,
                                                                 src/properties/sol_shrinking_box.rs:600-632
There is no CPU fallback here: if libmdx.so is missing or no GPU is usable this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from ._abi import (CCommDiag, CHBond, CConfig, CEnergies, CStats, CSystem, FORCE, MDX_EDEVICE, MDX_ENAN, MDX_EOOM,
                   MDX_EPARAM, MDX_OK, POS, VEL, MdConfig, MdSystem)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MDX_LIB") or os.path.join(_HERE, "libmdx.so")   # MDX_LIB: A/B builds of the same ABI


class ParamError(ValueError):
    """The reference's `ParamError { descrip }` (src/md/mod.rs:967)."""


class DeviceError(RuntimeError):
    """HIP/device failure (MDX_EDEVICE / MDX_EOOM); the reference degrades to its CPU path here
    (src/util.rs:1072-1119) — this library has none and says so loudly."""


class BlowUpError(FloatingPointError):
    """Non-finite state (MDX_ENAN); cf. the stop criteria at sol_shrinking_box.rs:776-789."""


_lib = None
_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)


def load_library():
    """Loads libmdx.so (built in-tree by __graft_entry__.build()).  Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (no CPU fallback exists).")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (+ HSA runtime).  If
    # libmdx.so were loaded first it would pull the system copy by RUNPATH and a later torch.cuda
    # initialisation then finds "No HIP GPUs" — and device pointers could not be shared with
    # torch.distributed buffers.  Importing torch first makes the dynamic linker resolve libmdx's
    # libamdhip64.so.7 dependency to the copy torch already mapped (same SONAME).
    try:
        import torch  # noqa: F401
    except ImportError:  # a host without PyTorch (e.g. the Rust application) simply uses the system runtime
        pass
    lib = C.CDLL(LIB_PATH)
    H = C.c_void_p
    lib.mdx_device_count.restype = C.c_int
    lib.mdx_last_error.restype = C.c_char_p
    lib.mdx_config_default.argtypes = [C.POINTER(CConfig)]
    lib.mdx_create.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), C.c_int, C.POINTER(H)]
    lib.mdx_destroy.argtypes = [H]
    lib.mdx_destroy.restype = None
    lib.mdx_step.argtypes = [H, C.c_float, _fp, C.c_uint32]
    lib.mdx_energy.argtypes = [H, C.POINTER(CEnergies)]
    lib.mdx_single_point.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), C.c_int,
                                     C.POINTER(CEnergies), _fp]
    lib.mdx_download.argtypes = [H, C.c_int, _fp]
    lib.mdx_upload.argtypes = [H, C.c_int, _fp]
    lib.mdx_upload_range.argtypes = [H, C.c_int, C.c_uint32, C.c_uint32, _fp]
    lib.mdx_set_box.argtypes = [H, C.c_float * 3, C.c_float * 3]
    lib.mdx_shrink_cell_towards.argtypes = [H, C.c_float * 3, C.c_float * 3, C.c_float, C.POINTER(C.c_int)]
    lib.mdx_rebuild_spatial_caches.argtypes = [H]
    lib.mdx_step_count.argtypes = [H]
    lib.mdx_step_count.restype = C.c_uint64
    lib.mdx_neighbor_list.argtypes = [H, _u32p, _u32p]
    lib.mdx_profile.argtypes = [H, C.c_int]
    lib.mdx_get_stats.argtypes = [H, C.POINTER(CStats)]
    lib.mdx_get_skin.argtypes = [H, _fp, C.POINTER(C.c_int)]
    lib.mdx_pair_launch_info.argtypes = [H, _u32p]
    lib.mdx_minimize_energy.argtypes = [H, C.c_uint32, _fp, C.c_float, C.POINTER(CEnergies), _u32p]
    lib.mdx_initialize_velocities.argtypes = [H, C.c_float, C.c_int, C.c_uint64]
    lib.mdx_set_thermostat.argtypes = [H, C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_uint64]
    lib.mdx_set_barostat.argtypes = [H, C.c_int, C.c_float, C.c_float, C.c_float, C.c_uint32]
    lib.mdx_set_integrator.argtypes = [H, C.c_int, C.c_float, C.c_float, C.c_uint64]
    lib.mdx_configure_alchemical_window.argtypes = [H, C.c_uint32, C.c_double]
    lib.mdx_set_alchemical_softcore.argtypes = [H, C.c_float, C.c_float]
    lib.mdx_get_box.argtypes = [H, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.mdx_set_zero_com_drift.argtypes = [H, C.c_int]
    lib.mdx_set_snapshot_cadence.argtypes = [H, C.c_uint32, C.c_int]
    lib.mdx_set_energy_cadence.argtypes = [H, C.c_uint32]
    lib.mdx_snapshot_count.argtypes = [H]
    lib.mdx_snapshot_count.restype = C.c_uint32
    lib.mdx_snapshot_read.argtypes = [H, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64),
                                      C.POINTER(CEnergies), _fp, _fp]
    lib.mdx_flush_snapshot_queues.argtypes = [H]
    lib.mdx_set_snapshot_handlers.argtypes = [H, _u32p]
    lib.mdx_snapshot_handler_mask.argtypes = [H, C.c_uint32]
    lib.mdx_snapshot_handler_mask.restype = C.c_uint32
    lib.mdx_snapshot_read_forces.argtypes = [H, C.c_uint32, _fp]
    lib.mdx_set_water_layout.argtypes = [H, C.c_uint32, C.c_uint32, C.c_uint32]
    lib.mdx_water_download.argtypes = [H, C.c_int, _fp, _fp, _fp, _fp]
    lib.mdx_set_hbond_detection.argtypes = [H, C.c_void_p, C.c_float, C.c_float]
    lib.mdx_snapshot_read_water.argtypes = [H, C.c_uint32, _fp, _fp, _fp]
    lib.mdx_snapshot_hbond_count.argtypes = [H, C.c_uint32]
    lib.mdx_snapshot_hbond_count.restype = C.c_uint32
    lib.mdx_snapshot_read_hbonds.argtypes = [H, C.c_uint32, C.POINTER(CHBond), C.c_uint32]
    lib.mdx_time_ps.argtypes = [H]
    lib.mdx_time_ps.restype = C.c_double
    lib.mdx_set_local_atoms.argtypes = [H, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_float * 3, C.c_float * 3, C.c_int32]
    lib.mdx_local_state.argtypes = [H, C.c_void_p, C.c_void_p]
    lib.mdx_chunk_begin.argtypes = [H]
    lib.mdx_chunk_integrate.argtypes = [H, C.c_int, C.c_float, C.c_uint32]
    lib.mdx_chunk_forces.argtypes = [H, C.c_int32]
    lib.mdx_chunk_end.argtypes = [H, C.c_uint32, _u32p]
    lib.mdx_flag_words.argtypes = [H]
    lib.mdx_flag_words.restype = C.c_void_p
    lib.mdx_stale_threshold.argtypes = [H]
    lib.mdx_stale_threshold.restype = C.c_uint32
    lib.mdx_add_steps.argtypes = [H, C.c_uint32]
    lib.mdx_pack_positions.argtypes = [H, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int32]
    lib.mdx_unpack_positions.argtypes = [H, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_int32]
    lib.mdx_stream.argtypes = [H]
    lib.mdx_stream.restype = C.c_void_p
    lib.mdx_comm_unique_id.argtypes = [C.c_char_p]
    lib.mdx_comm_init.argtypes = [H, C.c_char_p, C.c_int, C.c_int]
    lib.mdx_fabric_create.argtypes = [C.c_int]
    lib.mdx_fabric_create.restype = C.c_void_p
    lib.mdx_fabric_destroy.argtypes = [C.c_void_p]
    lib.mdx_fabric_destroy.restype = None
    lib.mdx_fabric_abort.argtypes = [C.c_void_p]
    lib.mdx_fabric_abort.restype = None
    lib.mdx_comm_init_fabric.argtypes = [H, C.c_void_p, C.c_int]
    lib.mdx_comm_init_null.argtypes = [H, C.c_int, C.c_int]
    lib.mdx_comm_init_shm.argtypes = [H, C.c_char_p, C.c_int, C.c_int]
    lib.mdx_comm_selftest.argtypes = [H]
    lib.mdx_comm_selftest_fault.argtypes = [H]
    lib.mdx_pme_info.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.mdx_pme_brick_overflows.argtypes = [H, C.POINTER(C.c_uint64)]
    lib.mdx_comm_debug_partition.argtypes = [H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _u32p, _u32p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.mdx_set_hydrogen_constraint.argtypes = [H, C.c_int, C.c_uint32, C.c_uint32, C.c_float]
    lib.mdx_constraint_description.argtypes = [H]
    lib.mdx_constraint_description.restype = C.c_char_p
    lib.mdx_comm_diag_read.argtypes = [H, C.POINTER(CCommDiag)]
    lib.mdx_set_energy_groups.argtypes = [H, C.c_void_p, C.c_uint32]
    lib.mdx_energy_group_count.argtypes = [H]
    lib.mdx_energy_group_count.restype = C.c_uint32
    lib.mdx_energy_between_mols.argtypes = [H, _fp, C.c_uint32]
    lib.mdx_snapshot_read_between_mols.argtypes = [H, C.c_uint32, _fp, C.c_uint32]
    lib.mdx_single_point_between_mols.argtypes = [C.POINTER(CSystem), C.POINTER(CConfig), C.c_int, C.c_void_p, C.c_uint32,
                                                  C.POINTER(CEnergies), _fp, _fp]
    lib.mdx_comm_info.argtypes = [H, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _u32p, _u32p, _fp]
    _lib = lib
    return lib


def _check(rc: int):
    if rc == MDX_OK:
        return
    msg = load_library().mdx_last_error().decode("utf-8", "replace")
    if rc == MDX_EPARAM:
        raise ParamError(msg)
    if rc == MDX_ENAN:
        raise BlowUpError(msg)
    if rc in (MDX_EDEVICE, MDX_EOOM):
        raise DeviceError(msg)
    raise RuntimeError(f"mdx error {rc}: {msg}")


def device_count() -> int:
    """`get_computation_device` probe (src/util.rs:1072-1119): 0 means "stay on the CPU path"."""
    return int(load_library().mdx_device_count())


class MdState:
    """`dynamics::MdState` as Molchanica uses it."""

    def __init__(self, system: MdSystem, cfg: MdConfig | None = None, device: int = 0):
        lib = load_library()
        self.cfg = cfg or MdConfig()
        self.system = system.normalise()
        self._h = C.c_void_p()
        cs, cc = self.system.to_c(), self.cfg.to_c()
        _check(lib.mdx_create(C.byref(cs), C.byref(cc), int(device), C.byref(self._h)))
        self.n_atoms = system.n_atoms

    # `MdState::new` spelling of the reference
    @classmethod
    def new(cls, system: MdSystem, cfg: MdConfig | None = None, device: int = 0) -> "MdState":
        return cls(system, cfg, device)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            load_library().mdx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- stepping --------------------------------------------------------------------------
    def step(self, dt: float, external_forces=None, n_steps: int = 1):
        """`md.step(&dev, dt, ext)`; n_steps > 1 keeps the GUI's 10-step burst
        (src/md/mod.rs:737) or a blocking run on the device."""
        ext = None
        if external_forces is not None:
            ext = np.ascontiguousarray(external_forces, dtype=np.float32).reshape(self.n_atoms, 3)
        _check(load_library().mdx_step(self._h, float(dt),
                                       None if ext is None else ext.ctypes.data_as(_fp), int(n_steps)))

    @property
    def step_count(self) -> int:
        return int(load_library().mdx_step_count(self._h))

    # -- energies / forces -------------------------------------------------------------------
    def energy(self) -> dict:
        """Per-snapshot `SnapshotEnergyData` superset (src/ui/panels/md_viewer.rs:195-257)."""
        e = CEnergies()
        _check(load_library().mdx_energy(self._h, C.byref(e)))
        return e.as_dict()

    def set_energy_groups(self, group_of_atom=None, n_groups: int = 0) -> int:
        """Groups of `energy_potential_between_mols` (src/properties/crystal.rs:533): None = one per molecule (mol_start);
        else group_of_atom[i] < n_groups <= 255.  -> number of groups in force (0: off)."""
        lib = load_library()
        if group_of_atom is None:
            _check(lib.mdx_set_energy_groups(self._h, None, 0))
        else:
            g = np.ascontiguousarray(group_of_atom, dtype=np.uint8).reshape(self.n_atoms)
            _check(lib.mdx_set_energy_groups(self._h, g.ctypes.data, int(n_groups)))
        return int(lib.mdx_energy_group_count(self._h))

    def clear_energy_groups(self):
        """Matrix output off again (`mdx_set_energy_groups(h, NULL, MDX_GROUPS_OFF)`): snapshots stop paying the extra pass."""
        _check(load_library().mdx_set_energy_groups(self._h, None, 0xFFFFFFFF))

    def energy_between_mols(self) -> np.ndarray:
        """`SnapshotEnergyData.energy_potential_between_mols` of the current state: symmetric [n, n] f32, kcal/mol."""
        lib = load_library()
        n = int(lib.mdx_energy_group_count(self._h))
        out = np.zeros((max(n, 1), max(n, 1)), dtype=np.float32)
        _check(lib.mdx_energy_between_mols(self._h, out.ctypes.data_as(_fp), n))
        return out

    def _download(self, which: int) -> np.ndarray:
        out = np.empty((self.n_atoms, 3), dtype=np.float32)
        _check(load_library().mdx_download(self._h, which, out.ctypes.data_as(_fp)))
        return out

    def positions(self) -> np.ndarray:
        return self._download(POS)

    def velocities(self) -> np.ndarray:
        return self._download(VEL)

    def forces(self) -> np.ndarray:
        return self._download(FORCE)

    def set_positions(self, pos):
        a = np.ascontiguousarray(pos, dtype=np.float32).reshape(self.n_atoms, 3)
        _check(load_library().mdx_upload(self._h, POS, a.ctypes.data_as(_fp)))

    def set_positions_range(self, first: int, pos):
        """New coordinates for atoms [first, first + len(pos)) - the docking loop's pose update
        (src/docking/mod.rs:81-154): the Verlet list is kept while the moved atoms stay inside its skin."""
        a = np.ascontiguousarray(pos, dtype=np.float32).reshape(-1, 3)
        _check(load_library().mdx_upload_range(self._h, POS, int(first), a.shape[0], a.ctypes.data_as(_fp)))

    def set_velocities(self, vel):
        a = np.ascontiguousarray(vel, dtype=np.float32).reshape(self.n_atoms, 3)
        _check(load_library().mdx_upload(self._h, VEL, a.ctypes.data_as(_fp)))

    # -- cell / spatial caches -----------------------------------------------------------------
    def set_cell(self, lo, hi):
        """`md.cell = SimBox::new(lo, hi)` (sol_shrinking_box.rs:600-603)."""
        _check(load_library().mdx_set_box(self._h, (C.c_float * 3)(*map(float, lo)),
                                          (C.c_float * 3)(*map(float, hi))))

    def shrink_cell_towards(self, target_lo, target_hi, shrink_per_step: float) -> bool:
        """`md.shrink_cell_towards(dev, target_cell, cfg) -> bool` (src/properties/sol_shrinking_box.rs:990)."""
        lo = (C.c_float * 3)(*[float(v) for v in target_lo]); hi = (C.c_float * 3)(*[float(v) for v in target_hi])
        out = C.c_int(0)
        _check(load_library().mdx_shrink_cell_towards(self._h, lo, hi, float(shrink_per_step), C.byref(out)))
        return bool(out.value)

    def rebuild_spatial_caches(self):
        _check(load_library().mdx_rebuild_spatial_caches(self._h))

    def neighbor_list(self):
        """-> (offsets [N+1], idx) of the Verlet list at the last rebuild, caller atom order."""
        lib = load_library()
        off = np.zeros(self.n_atoms + 1, dtype=np.uint32)
        _check(lib.mdx_neighbor_list(self._h, off.ctypes.data_as(_u32p), None))
        idx = np.zeros(max(int(off[-1]), 1), dtype=np.uint32)
        _check(lib.mdx_neighbor_list(self._h, off.ctypes.data_as(_u32p), idx.ctypes.data_as(_u32p)))
        return off, idx[: int(off[-1])]

    # -- callers either side of step -------------------------------------------------
    def minimize_energy(self, max_iters: int, external_forces=None, f_tol: float = 0.0):
        """`md.minimize_energy(dev, iters, ext)` (src/ui/mol_editor.rs:375).  -> (energies, iterations)."""
        ext = None
        if external_forces is not None:
            ext = np.ascontiguousarray(external_forces, dtype=np.float32).reshape(self.n_atoms, 3)
        e, it = CEnergies(), C.c_uint32(0)
        _check(load_library().mdx_minimize_energy(self._h, int(max_iters), None if ext is None else ext.ctypes.data_as(_fp),
                                                  float(f_tol), C.byref(e), C.byref(it)))
        return e.as_dict(), int(it.value)

    def initialize_velocities(self, temperature: float, zero_com_drift: bool = True, seed: int = 0):
        """`md.initialize_velocities(TEMP, zero_com_drift)` (sol_shrinking_box.rs:965)."""
        _check(load_library().mdx_initialize_velocities(self._h, float(temperature), int(zero_com_drift), int(seed)))

    def set_thermostat(self, kind: int, temp_target: float, tau_ps: float, every_n_steps: int = 10, seed: int = 0):
        """`Integrator::VerletVelocity{thermostat: Some(tau)}` + `temp_target`; kind 1 Berendsen, 2 CSVR."""
        _check(load_library().mdx_set_thermostat(self._h, int(kind), float(temp_target), float(tau_ps),
                                                 int(every_n_steps), int(seed)))

    def configure_alchemical_window(self, mol_index: int, lam: float):
        """`md.configure_alchemical_window(dev, mol_index, lambda)` (src/properties/water_sol.rs:556); lam < 0 = off."""
        _check(load_library().mdx_configure_alchemical_window(self._h, int(mol_index), float(lam)))

    def set_alchemical_softcore(self, alpha: float = 0.5, sigma_min: float = 3.0):
        """Soft core of the alchemical window (alpha = 0: linear coupling)."""
        _check(load_library().mdx_set_alchemical_softcore(self._h, float(alpha), float(sigma_min)))

    def set_integrator(self, kind: int, gamma_per_ps: float = 1.0, temperature: float = 300.0, seed: int = 0):
        """`Integrator::{VerletVelocity (0), Leapfrog (1), LangevinMiddle{gamma} (2)}` (md.rs:296-305)."""
        _check(load_library().mdx_set_integrator(self._h, int(kind), float(gamma_per_ps), float(temperature), int(seed)))

    def set_barostat(self, kind: int, pressure_target_bar: float = 1.0, tau_ps: float = 5.0,
                     compressibility_per_bar: float = 4.5e-5, every_n_steps: int = 25):
        """`MdConfig.barostat_cfg = Some(BarostatCfg{tau, pressure_target})` (md.rs:517-557); kind 1 Berendsen."""
        _check(load_library().mdx_set_barostat(self._h, int(kind), float(pressure_target_bar), float(tau_ps),
                                               float(compressibility_per_bar), int(every_n_steps)))

    def cell(self):
        """`md.cell` -> (bounds_low, bounds_high)."""
        lo = (C.c_float * 3)(); hi = (C.c_float * 3)()
        _check(load_library().mdx_get_box(self._h, lo, hi))
        return np.array(lo[:], np.float32), np.array(hi[:], np.float32)

    def set_zero_com_drift(self, enable: bool = True):
        _check(load_library().mdx_set_zero_com_drift(self._h, int(enable)))

    def set_energy_cadence(self, every_n: int):
        """The caller will read `energy()` after every `every_n`-th step (the ratio of the reference's snapshot handlers,
        src/md/mod.rs:121-122): the step loop then evaluates the energies with the forces of those steps."""
        _check(load_library().mdx_set_energy_cadence(self._h, int(every_n)))

    def set_snapshot_cadence(self, every_n: int, with_velocities: bool = False):
        """`snapshot_handlers.memory: Some(every_n)` (water_sol.rs:185-189)."""
        _check(load_library().mdx_set_snapshot_cadence(self._h, int(every_n), int(with_velocities)))

    SNAP_HANDLERS = ("memory", "dcd", "nstxout", "nstvout", "nstfout", "nstenergy", "nstcalcenergy", "nstxout_compressed")

    def set_snapshot_handlers(self, **every_n):
        """`MdConfig.snapshot_handlers {memory, dcd, gromacs{nstxout, nstvout, nstfout, nstenergy, nstcalcenergy, nstxout_compressed}}`
        (src/properties/crystal.rs:335-342): a cadence per handler; every snapshot says which handlers wanted it."""
        a = np.array([int(every_n.pop(k, 0) or 0) for k in self.SNAP_HANDLERS], dtype=np.uint32)
        if every_n:
            raise ParamError("unknown snapshot handler: " + ", ".join(every_n))
        _check(load_library().mdx_set_snapshot_handlers(self._h, a.ctypes.data_as(_u32p)))

    @property
    def snapshots(self) -> list:
        """`md.snapshots` (src/md/mod.rs:121-122): list of dicts time/step/energy_data/atom_posits[/velocities]."""
        lib = load_library()
        out = []
        for k in range(int(lib.mdx_snapshot_count(self._h))):
            t, st, e = C.c_double(), C.c_uint64(), CEnergies()
            pos = np.empty((self.n_atoms, 3), dtype=np.float32)
            vel = np.empty((self.n_atoms, 3), dtype=np.float32)
            rc = lib.mdx_snapshot_read(self._h, k, C.byref(t), C.byref(st), C.byref(e), pos.ctypes.data_as(_fp),
                                       vel.ctypes.data_as(_fp))
            if rc != MDX_OK:   # taken without velocities
                _check(lib.mdx_snapshot_read(self._h, k, C.byref(t), C.byref(st), C.byref(e), pos.ctypes.data_as(_fp), None))
                vel = None
            snap = dict(time=float(t.value), step=int(st.value), energy_data=e.as_dict(), atom_posits=pos,
                        atom_velocities=vel)
            if self._n_waters:      # the reference's Snapshot: non-water atoms + water_o / h0 / h1 (src/md/viewer.rs:374-394)
                o, h0, h1 = (np.empty((self._n_waters, 3), dtype=np.float32) for _ in range(3))
                _check(lib.mdx_snapshot_read_water(self._h, k, o.ctypes.data_as(_fp), h0.ctypes.data_as(_fp), h1.ctypes.data_as(_fp)))
                snap.update(all_posits=pos, atom_posits=pos[:self._water_first], water_o_posits=o, water_h0_posits=h0, water_h1_posits=h1)
            mask = int(lib.mdx_snapshot_handler_mask(self._h, k))
            snap["handlers"] = [n for i, n in enumerate(self.SNAP_HANDLERS) if mask >> i & 1]
            if mask >> 4 & 1:
                frc = np.empty((self.n_atoms, 3), dtype=np.float32)
                if lib.mdx_snapshot_read_forces(self._h, k, frc.ctypes.data_as(_fp)) == MDX_OK:
                    snap["atom_forces"] = frc
            n_g = int(lib.mdx_energy_group_count(self._h))
            if n_g:
                m = np.zeros((n_g, n_g), dtype=np.float32)
                if lib.mdx_snapshot_read_between_mols(self._h, k, m.ctypes.data_as(_fp), n_g) == MDX_OK:
                    snap["energy_data"]["energy_potential_between_mols"] = m
            n_hb = int(lib.mdx_snapshot_hbond_count(self._h, k))
            hb = (CHBond * max(n_hb, 1))()
            if n_hb:
                _check(lib.mdx_snapshot_read_hbonds(self._h, k, hb, n_hb))
            snap["energy_data"]["hydrogen_bonds"] = [
                dict(donor=(b.donor_type, b.donor), acceptor=(b.acceptor_type, b.acceptor), hydrogen=(b.hydrogen_type, b.hydrogen),
                     strength=float(b.strength)) for b in hb[:n_hb]]
            out.append(snap)
        return out

    # -- md.water / hydrogen bonds ---------------------------------------------------------------
    _n_waters = 0
    _water_first = 0
    _water_sites = 0

    def set_water_layout(self, first_atom: int, n_waters: int, sites_per_water: int):
        """Where the solvent waters sit in the flat atom array: the reference keeps them apart in `md.water`
        (src/properties/sol_shrinking_box.rs:605-613)."""
        _check(load_library().mdx_set_water_layout(self._h, int(first_atom), int(n_waters), int(sites_per_water)))
        self._water_first, self._n_waters, self._water_sites = int(first_atom), int(n_waters), int(sites_per_water)

    def water(self, which: str = "posit") -> dict:
        """`md.water[i].{o,h0,h1,m}.posit / .force` as arrays [n_waters, 3] (sol_shrinking_box.rs:780-786)."""
        n = self._n_waters
        arr = {k: np.empty((n, 3), dtype=np.float32) for k in (("o", "h0", "h1", "m") if self._water_sites == 4 else ("o", "h0", "h1"))}
        _check(load_library().mdx_water_download(self._h, POS if which == "posit" else FORCE, arr["o"].ctypes.data_as(_fp),
                                                 arr["h0"].ctypes.data_as(_fp), arr["h1"].ctypes.data_as(_fp),
                                                 arr["m"].ctypes.data_as(_fp) if "m" in arr else None))
        return arr

    def set_hbond_detection(self, is_heavy_nosf, max_h_acc_dist: float = 2.5, min_angle_deg: float = 120.0):
        """Hydrogen bonds in every snapshot's energy data (src/md/viewer.rs:917-960); None switches it off."""
        if is_heavy_nosf is None:
            _check(load_library().mdx_set_hbond_detection(self._h, None, 0.0, 0.0))
            return
        m = np.ascontiguousarray(is_heavy_nosf, dtype=np.uint8).reshape(self.n_atoms)
        _check(load_library().mdx_set_hbond_detection(self._h, m.ctypes.data, float(max_h_acc_dist), float(min_angle_deg)))

    def flush_snapshot_queues(self):
        _check(load_library().mdx_flush_snapshot_queues(self._h))

    @property
    def time_ps(self) -> float:
        return float(load_library().mdx_time_ps(self._h))

    # -- profiling ---------------------------------------------------------------------------
    def profile(self, enable=True):
        """0/False off, 1/True every step kernel, 2 the pair kernel only, 3 (decomposed handles) every phase of the step in its
        production arrangement (comm_diag reads the sums)."""
        _check(load_library().mdx_profile(self._h, int(enable)))

    def stats(self) -> dict:
        s = CStats()
        _check(load_library().mdx_get_stats(self._h, C.byref(s)))
        return s.as_dict()

    def pair_launch_info(self) -> dict:
        """Which pair-kernel instantiation ran last: {"step": {...}, "any": {...}} (include/mdx.h: mdx_pair_launch_info)."""
        out = (C.c_uint32 * 24)()
        _check(load_library().mdx_pair_launch_info(self._h, out))
        keys = ("waves_per_tile", "dual", "half", "coulomb", "energy", "workgroups_per_tile", "bonded_workgroups", "tiles")
        return {"step": dict(zip(keys, (int(v) for v in out[0:8]))), "any": dict(zip(keys, (int(v) for v in out[8:16]))),
                "inner_lists_from_rebuilds": int(out[16]), "last_rebuild_wrote_the_inner_list": bool(out[17]),
                "water_step_launches": int(out[18]), "water_step_mixed_launches": int(out[19]),
                "one_launch_steps": int(out[20]), "kicks_beyond_grant": int(out[21])}

    def skin(self):
        """-> (Verlet skin in force, still tuning?)  (MdConfig.skin == 0 lets the library choose it)."""
        v, t = C.c_float(), C.c_int()
        _check(load_library().mdx_get_skin(self._h, C.byref(v), C.byref(t)))
        return float(v.value), bool(t.value)

    def computation_time(self) -> float:
        """`md.computation_time()` (src/md/mod.rs:740-743): ms spent inside step calls."""
        return float(self.stats()["wall_ms_sum"])

    # -- multi-GPU: one box decomposed over the ranks, all below the C ABI (include/mdx.h) --------------
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """Join the RCCL communicator (ncclCommInitRank) and take this rank's share of the box.  From here on
        `step`, `energy`, `positions` / `velocities` / `forces` are COLLECTIVE calls."""
        assert len(unique_id) == 128
        _check(load_library().mdx_comm_init(self._h, bytes(unique_id), int(rank), int(world)))

    def comm_init_fabric(self, fabric: "Fabric", rank: int):
        """The same decomposition between handles of ONE process (one thread per rank) through an in-process fabric."""
        self._fabric = fabric            # keep it alive as long as the handle
        _check(load_library().mdx_comm_init_fabric(self._h, fabric.ptr, int(rank)))

    def comm_init_shm(self, name: str, rank: int, world: int):
        """Ranks as processes of one host, rows staged through POSIX shared memory `/mdx_<name>` (no RCCL)."""
        _check(load_library().mdx_comm_init_shm(self._h, name.encode(), int(rank), int(world)))

    def comm_init_null(self, rank: int, world: int):
        """Rank `rank` of `world` with a transport that delivers nothing (one rank's cost measured alone)."""
        _check(load_library().mdx_comm_init_null(self._h, int(rank), int(world)))

    def comm_selftest(self):
        """Every transport entry point on the real wire, results checked (collective)."""
        _check(load_library().mdx_comm_selftest(self._h))

    def comm_selftest_fault(self):
        """Forces a failing call inside a send/recv group and checks that the transport reports it, closes the group and
        refuses further traffic.  The communicator is unusable afterwards: close the handle."""
        _check(load_library().mdx_comm_selftest_fault(self._h))

    def comm_debug_partition(self, n_atoms: int) -> dict:
        """What the partition kernels derived at the last (re)partition (diagnostics; see include/mdx.h)."""
        cls = np.zeros(n_atoms, np.uint8); owner = np.zeros(n_atoms, np.uint8); code = np.zeros(n_atoms, np.uint8)
        mask = np.zeros(n_atoms, np.uint32)
        ns, nr = C.c_uint32(), C.c_uint32()
        lib = load_library()
        _check(lib.mdx_comm_debug_partition(self._h, cls.ctypes.data, owner.ctypes.data, code.ctypes.data, mask.ctypes.data,
                                            C.byref(ns), C.byref(nr), None, None, 0))
        sid = np.zeros(max(ns.value, 1), np.uint32); rid = np.zeros(max(nr.value, 1), np.uint32)
        _check(lib.mdx_comm_debug_partition(self._h, None, None, None, None, C.byref(ns), C.byref(nr), sid.ctypes.data, rid.ctypes.data,
                                            max(ns.value, nr.value, 1)))
        return dict(cls=cls, owner=owner, image_code=code, send_mask=mask, send_ids=sid[:ns.value], recv_ids=rid[:nr.value])

    def pme_info(self) -> dict:
        """Is the reciprocal-space mesh of this (decomposed) handle slab-decomposed, and what it sends per force call."""
        on, a, b, c = C.c_int(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(load_library().mdx_pme_info(self._h, C.byref(on), C.byref(a), C.byref(b), C.byref(c)))
        return dict(slab_on=bool(on.value), mesh_bytes_sent=a.value, transpose_bytes_sent=b.value, replicated_mesh_bytes=c.value)

    def pme_brick_overflows(self) -> int:
        """Atoms (summed over all force calls) that went through the overflow list of the brick spread."""
        n = C.c_uint64()
        _check(load_library().mdx_pme_brick_overflows(self._h, C.byref(n)))
        return int(n.value)

    def set_hydrogen_constraint(self, kind: str, order: int = 4, iters: int = 1, shake_tolerance: float = 0.0) -> str:
        """HydrogenConstraint::{Flexible, Shake{shake_tolerance}, Linear{order, iter}} (src/ui/panels/md.rs:362-371) -> the
        solver's description of what it will do (how Linear was mapped)."""
        k = {"flexible": 0, "shake": 1, "linear": 2}[kind.lower()]
        lib = load_library()
        _check(lib.mdx_set_hydrogen_constraint(self._h, k, int(order), int(iters), float(shake_tolerance)))
        return lib.mdx_constraint_description(self._h).decode()

    def comm_diag(self) -> dict:
        """mdx_comm_diag_read: transport, RCCL version / communicator size, owned / ghost atoms, halo bytes per step, repartitions,
        whether the interior / boundary split was kept, and - summed while profile(3) is on - the GPU time of every phase of the
        decomposed step."""
        d = CCommDiag()
        _check(load_library().mdx_comm_diag_read(self._h, C.byref(d)))
        return d.as_dict()

    def comm_info(self) -> dict:
        r, w, g = C.c_int(), C.c_int(), (C.c_int * 3)()
        no, ng, halo = C.c_uint32(), C.c_uint32(), C.c_float()
        _check(load_library().mdx_comm_info(self._h, C.byref(r), C.byref(w), g, C.byref(no), C.byref(ng), C.byref(halo)))
        return dict(rank=r.value, world=w.value, grid=tuple(g), n_owned=no.value, n_ghost=ng.value, halo=halo.value)

    # -- multi-GPU plumbing (raw device pointers; the building blocks for hosts that drive a decomposition themselves; tests/decomp_spec.py does) -------------------
    def set_local_atoms(self, n_local, d_gid, d_ghost, d_pos4, d_vel4, lo, hi, periodic_mask: int):
        _check(load_library().mdx_set_local_atoms(
            self._h, int(n_local), C.c_void_p(d_gid), C.c_void_p(d_ghost), C.c_void_p(d_pos4), C.c_void_p(d_vel4),
            (C.c_float * 3)(*map(float, lo)), (C.c_float * 3)(*map(float, hi)), int(periodic_mask)))

    def local_state(self, d_pos4, d_vel4):
        _check(load_library().mdx_local_state(self._h, C.c_void_p(d_pos4), C.c_void_p(d_vel4)))

    def chunk_begin(self):
        _check(load_library().mdx_chunk_begin(self._h))

    def chunk_integrate(self, mode: int, dt: float, s: int):
        _check(load_library().mdx_chunk_integrate(self._h, int(mode), float(dt), int(s)))

    def chunk_forces(self, s: int):
        _check(load_library().mdx_chunk_forces(self._h, int(s)))

    def chunk_end(self, n_words: int) -> np.ndarray:
        out = np.zeros(n_words, dtype=np.uint32)
        _check(load_library().mdx_chunk_end(self._h, int(n_words), out.ctypes.data_as(_u32p)))
        return out

    def flag_words_ptr(self) -> int:
        return int(load_library().mdx_flag_words(self._h) or 0)

    def stale_threshold(self) -> int:
        return int(load_library().mdx_stale_threshold(self._h))

    def add_steps(self, n: int):
        _check(load_library().mdx_add_steps(self._h, int(n)))

    def pack_positions(self, d_gid: int, n: int, d_out4: int, flag_word: int = -1):
        _check(load_library().mdx_pack_positions(self._h, C.c_void_p(d_gid), int(n), C.c_void_p(d_out4), int(flag_word)))

    def unpack_positions(self, d_gid: int, n: int, d_in4: int, d_shift4: int = 0, flag_word: int = -1):
        _check(load_library().mdx_unpack_positions(self._h, C.c_void_p(d_gid), int(n), C.c_void_p(d_in4),
                                                   C.c_void_p(d_shift4) if d_shift4 else None, int(flag_word)))

    def stream_ptr(self) -> int:
        return int(load_library().mdx_stream(self._h) or 0)


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 calls it and hands the 128 bytes to the other ranks)."""
    buf = C.create_string_buffer(128)
    _check(load_library().mdx_comm_unique_id(buf))
    return buf.raw


class Fabric:
    """In-process meeting point of `world` decomposed handles (mdx_fabric_create)."""

    def __init__(self, world: int):
        self.world = int(world)
        self.ptr = load_library().mdx_fabric_create(self.world)
        if not self.ptr:
            raise ParamError("mdx_fabric_create failed")

    def abort(self):
        if self.ptr:
            load_library().mdx_fabric_abort(self.ptr)

    def __del__(self):
        try:
            if self.ptr:
                load_library().mdx_fabric_destroy(self.ptr)
                self.ptr = None
        except Exception:
            pass


def run_dynamics_blocking(md: MdState, dt: float, n_steps: int):
    """`run_dynamics_blocking` (src/md/mod.rs:696-724): n plain steps, no early exit."""
    if n_steps == 0:
        return
    md.step(dt, None, n_steps)


def release_single_point_cache():
    """Frees the device state `compute_energy_snapshot` keeps between calls (mdx_single_point_release)."""
    lib = load_library()
    lib.mdx_single_point_release.restype = None
    lib.mdx_single_point_release()


def compute_energy_snapshot(system: MdSystem, cfg: MdConfig | None = None, device: int = 0,
                            with_forces: bool = False, groups=None, n_groups: int = 0):
    """`dynamics::compute_energy_snapshot` (src/md/mod.rs:1036): stateless single-point scorer.
    Returns the energy dict (and forces [N,3] when asked).  groups: "mol" or a [N] group map -> the dict also carries
    `energy_potential_between_mols` [n, n] (the receptor-ligand interaction energy of a docking pose is one element)."""
    lib = load_library()
    cfg = cfg or MdConfig()
    system.normalise()
    cs, cc = system.to_c(), cfg.to_c()
    e = CEnergies()
    f = np.zeros((system.n_atoms, 3), dtype=np.float32) if with_forces else None
    if groups is not None:
        by_mol = isinstance(groups, str)
        n = int(system.mol_start.size) if by_mol else int(n_groups)
        g = None if by_mol else np.ascontiguousarray(groups, dtype=np.uint8).reshape(system.n_atoms)
        m = np.zeros((n, n), dtype=np.float32)
        _check(lib.mdx_single_point_between_mols(C.byref(cs), C.byref(cc), int(device), None if g is None else g.ctypes.data, n,
                                                 C.byref(e), None if f is None else f.ctypes.data_as(_fp), m.ctypes.data_as(_fp)))
        d = e.as_dict(); d["energy_potential_between_mols"] = m
        return (d, f) if with_forces else d
    _check(lib.mdx_single_point(C.byref(cs), C.byref(cc), int(device), C.byref(e),
                                None if f is None else f.ctypes.data_as(_fp)))
    return (e.as_dict(), f) if with_forces else e.as_dict()

"""molchanica_amd — MI355X-native MD force engine behind Molchanica's `src/md` step loop.

Host-side mirror of the `dynamics` crate surface the reference consumes
(/root/reference src/md/mod.rs:13-16, 689, 716, 748, 1036) over the C ABI in include/mdx.h.
"""
from ._abi import (MdConfig, MdSystem, SimBoxInit, COULOMB_SHIFTED, COULOMB_REACTION, COULOMB_EWALD,
                   COMBINE_LORENTZ_BERTHELOT, COMBINE_GEOMETRIC, OVR_BONDED_DISABLED,
                   OVR_COULOMB_DISABLED, OVR_LJ_DISABLED, OVR_LONG_RANGE_RECIP_DISABLED,
                   ATOM_STATIC, ATOM_BONDED_ONLY, ATOM_GHOST, POS, VEL, FORCE)

__all__ = ["MdConfig", "MdSystem", "SimBoxInit", "MdState", "ParamError", "compute_energy_snapshot"]


def __getattr__(name):
    # MdState needs the HIP library; import lazily so CPU-only tooling can use the ABI types.
    if name in ("MdState", "ParamError", "compute_energy_snapshot", "device_count",
                "run_dynamics_blocking", "DeviceError"):
        from . import md_state
        return getattr(md_state, name)
    raise AttributeError(name)

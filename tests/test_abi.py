"""The C-ABI library loads and exports every symbol include/mdx.h declares; struct layouts of the
ctypes mirror match the C compiler's.  CPU only: no compute call is made without a GPU."""
import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from molchanica_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mdx.h")
LIB = os.path.join(ROOT, "molchanica_amd", "libmdx.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdx_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g
        g.build()
    return C.CDLL(LIB)


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("mdx_create", "mdx_destroy", "mdx_step", "mdx_energy", "mdx_single_point", "mdx_download",
                 "mdx_upload", "mdx_set_box", "mdx_rebuild_spatial_caches", "mdx_step_count",
                 "mdx_neighbor_list", "mdx_last_error", "mdx_device_count", "mdx_config_default"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"libmdx.so does not export {name}"


def test_every_export_is_bound_or_listed():
    """INTEGRATION.md is what a maintainer of the reference copies from: every export of include/mdx.h is either declared in its Rust
    `extern "C"` block or named in its "not bound" list, and neither names a symbol the header does not have."""
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blk = txt[txt.index('extern "C" {'):]
    blk = blk[:blk.index("\n}\n")]
    bound = set(re.findall(r"pub fn (mdx_[a-z0-9_]+)", blk))
    nb = txt[txt.index("<!-- not-bound:begin -->"):txt.index("<!-- not-bound:end -->")]
    listed = set(re.findall(r"`(mdx_[a-z0-9_]+)`", nb))
    syms = set(declared_symbols())
    assert not (syms - bound - listed), f"exports neither bound nor listed in INTEGRATION.md: {sorted(syms - bound - listed)}"
    assert not ((bound | listed) - syms), f"INTEGRATION.md names symbols include/mdx.h does not declare: {sorted((bound | listed) - syms)}"
    assert not (bound & listed)
    # md.computation_time() (/root/reference src/md/mod.rs:740-743) needs these two and the struct
    assert {"mdx_get_stats", "mdx_profile", "mdx_time_ps"} <= bound and "pub struct MdxStats" in txt


def test_rust_stats_struct_mirrors_the_ctypes_one():
    txt = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = txt[txt.index("pub struct MdxStats {"):]
    body = body[:body.index("\n}\n")]
    body = re.sub(r"//.*", "", body)
    rust = re.findall(r"pub ([a-z0-9_]+): (u64|u32|f64)", body)
    want = {C.c_uint64: "u64", C.c_uint32: "u32", C.c_double: "f64"}
    assert rust == [(n, want[t]) for n, t in _abi.CStats._fields_]


def test_struct_layouts_match_c(tmp_path):
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "mdx.h"
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(mdx_system), sizeof(mdx_config), sizeof(mdx_energies), sizeof(mdx_stats));
  printf("%zu %zu %zu %zu\n", offsetof(mdx_system, n_bonds), offsetof(mdx_system, excl_offsets),
         offsetof(mdx_system, periodic), offsetof(mdx_system, box_hi));
  printf("%zu %zu\n", offsetof(mdx_config, chunk_steps), offsetof(mdx_stats, nb_ms_sum));
  return 0; }
'''
    src = tmp_path / "t.c"
    src.write_text(prog)
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    vals = list(map(int, out))
    S, Cf, E, St = _abi.CSystem, _abi.CConfig, _abi.CEnergies, _abi.CStats
    assert vals[:4] == [C.sizeof(S), C.sizeof(Cf), C.sizeof(E), C.sizeof(St)]
    assert vals[4:8] == [S.n_bonds.offset, S.excl_offsets.offset, S.periodic.offset, S.box_hi.offset]
    assert vals[8:] == [Cf.chunk_steps.offset, St.nb_ms_sum.offset]


def test_config_default_and_error_string(lib):
    lib.mdx_config_default.argtypes = [C.POINTER(_abi.CConfig)]
    c = _abi.CConfig()
    lib.mdx_config_default(C.byref(c))
    assert c.lj_cutoff == 10.0 and c.coulomb_cutoff == 10.0 and c.skin == 2.0
    assert abs(c.coulomb_k - 332.0637) < 1e-3 and c.scale14_lj == 0.5
    assert abs(c.scale14_coulomb - 1 / 1.2) < 1e-6 and c.chunk_steps == 16
    d = _abi.MdConfig()
    assert (d.lj_cutoff, d.skin, d.scale14_lj, d.chunk_steps) == (c.lj_cutoff, c.skin, c.scale14_lj, c.chunk_steps)
    lib.mdx_last_error.restype = C.c_char_p
    assert isinstance(lib.mdx_last_error(), bytes)


def test_bad_input_is_rejected_before_touching_a_device(lib):
    """Parameter validation runs first, so it can be exercised on a CPU-only host."""
    from molchanica_amd import systems
    from molchanica_amd.md_state import MdState, ParamError
    s = systems.lig50()
    s.bond_idx = s.bond_idx.copy()
    s.bond_idx[0, 0] = 10_000
    with pytest.raises(ParamError, match="out of range"):
        MdState(s, _abi.MdConfig(lj_cutoff=0, coulomb_cutoff=0))
    s = systems.lig50()
    s.mass = s.mass.copy()
    s.mass[3] = 0.0
    with pytest.raises(ParamError, match="mass"):
        MdState(s, _abi.MdConfig(lj_cutoff=0, coulomb_cutoff=0))


def test_no_silent_cpu_fallback(lib):
    """Without a GPU, creating a state must fail loudly (DeviceError), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from molchanica_amd import systems
    from molchanica_amd.md_state import DeviceError, MdState, device_count
    assert device_count() == 0
    with pytest.raises(DeviceError):
        MdState(systems.lig50(), _abi.MdConfig(lj_cutoff=0, coulomb_cutoff=0))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "molchanica_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"(import\s+oracle|from\s+oracle|oracle[/.]|liborc|orc_)", txt), \
                    f"{f} reaches into the oracle"


def test_reference_kernels_build_from_where_they_lie():
    """oracle/_ref: the reference's own CUDA pair kernels compiled by hipcc from /root/reference (only where that tree exists)."""
    from oracle import ref_kernels
    if not os.path.isdir(ref_kernels.REFERENCE):
        pytest.skip("no reference tree on this machine: the prebuilt oracle/_ref/libref_cuda.so is what travels")
    assert ref_kernels.build() is not None
    l = C.CDLL(ref_kernels.PATH)
    for name in ("ref_lj_force", "ref_coulomb_force", "ref_lj_V", "ref_min_image"):
        assert hasattr(l, name)
    txt = open(os.path.join(ROOT, "oracle", "ref_cuda_host.hip")).read()
    assert '#include "cuda.cu"' in txt and "__global__ void lj_force_kernel" not in txt, "the reference's kernels are included, never copied"

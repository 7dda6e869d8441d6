"""The compiled-language host above the C ABI: include/mdx.hpp (C++17 mirror of the `MdState` surface the reference's
Rust host calls, /root/reference src/md/mod.rs:689-750) and tests/cpp/host_driver.cpp.

CPU: the header and the driver compile and link against libmdx.so with plain g++ (no HIP headers needed above the ABI).
GPU: the driver runs: creation, ParamError, minimiser, velocities, F = -dE/dx through the stateless scorer, NVE energy
conservation in 10-step bursts, snapshots, move semantics."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "molchanica_amd")
OUT = os.path.join(ROOT, "tests", "cpp", "build")
EXE = os.path.join(OUT, "host_driver")


def _build():
    os.makedirs(OUT, exist_ok=True)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "host_driver.cpp"), "-o", EXE,
           "-L", PKG, "-lmdx", "-pthread", f"-Wl,-rpath,{PKG}", "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return EXE


def test_cpp_host_compiles_and_links():
    assert os.path.exists(os.path.join(PKG, "libmdx.so")), "build the library first (__graft_entry__.build())"
    exe = _build()
    assert os.path.getsize(exe) > 0


@pytest.mark.gpu
def test_cpp_host_driver_runs():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ALL OK" in r.stdout

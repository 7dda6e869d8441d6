#!/usr/bin/env python3
"""Records the outputs of the REFERENCE's own pair kernels (oracle/_ref/libref_cuda.so = /root/reference/src/cuda/cuda.cu +
util.cu compiled as they lie) into tests/golden/ref_pair_kernels.npz, so that the pin they give survives a checkout without
/root/reference and without the prebuilt object.  Run on the GPU box (the kernels execute on the MI355X):

    gpurun -- 'python3 tests/golden/make_ref_pair_kernels.py gpurun_out/ref_pair_kernels.npz'   # then copy into tests/golden/

Inputs are the seeded cases of tests/ref_cases.py; only outputs are stored."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from oracle import ref_kernels as ref  # noqa: E402
from tests import ref_cases as rc  # noqa: E402


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else rc.FIXTURE
    assert ref.available(), "oracle/_ref/libref_cuda.so is missing: make -C oracle ref (needs /root/reference)"
    out = {}
    mi = rc.min_image_cases()
    out["min_image"] = np.stack([ref.min_image(e, d) for e, d in mi]).astype(np.float32)
    _, _, _, targets, cases = rc.dhfr_case()
    f_lj, f_c = rc.run_reference_on_dhfr(ref, cases)
    out["dhfr_targets"] = np.asarray(targets, np.int64)
    out["dhfr_n_src"] = np.array([len(c["nb"]) for c in cases], np.int64)
    out["dhfr_f_lj"] = f_lj.astype(np.float32); out["dhfr_f_coul_k1"] = f_c.astype(np.float32)
    # formula cases (tests/test_gpu_reference_kernels.py)
    a, b, rng = rc.two_groups(1)
    sig_t, eps_t = np.array([3.4, 3.0, 2.6]), np.array([0.10, 0.17, 0.05])
    ta, tb = rng.integers(0, 3, len(a)), rng.integers(0, 3, len(b))
    out["lj_force_seed1"] = ref.lj_force(a, b, 0.5 * (sig_t[ta][:, None] + sig_t[tb][None, :]), np.sqrt(eps_t[ta][:, None] * eps_t[tb][None, :]))
    a, b, rng = rc.two_groups(2, 48, 48)
    q = rng.normal(0, 0.4, 48)
    out["coulomb_force_seed2"] = ref.coulomb_force(a, b, q)
    a, b, _ = rc.two_groups(3)
    out["lj_V_seed3"] = ref.lj_V(b, a, 3.2, 0.12)
    os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()

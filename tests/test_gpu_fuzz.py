"""The randomised evidence inside the suite: tests/fuzz_parity.py (GPU vs oracle over random systems, cut-offs, skins, Coulomb modes,
kernel variants and alternate-path knobs) and tests/fuzz_decomp.py (2 / 4 / 8 virtual ranks vs one GPU, the plain shifted cut-off -
the mode `bench.py --gpus N` times - included) with fixed seeds.  Each runs as a process of its own: the knobs are environment
variables, several of them read once per process."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_script(name, cases, seed, extra_env=None):
    env = dict(os.environ)
    for k in [k for k in env if k.startswith("MDX_") or k.startswith("FUZZ_")]:
        env.pop(k)
    env.update(extra_env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", name), str(cases), str(seed)], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    tail = "\n".join((p.stdout + p.stderr).splitlines()[-25:])
    assert p.returncode == 0, tail
    m = re.search(r"(\d+) of (\d+) cases", p.stdout)
    assert m, tail
    fails = [l for l in p.stdout.splitlines() if l.startswith("FAIL")]
    assert int(m.group(1)) == int(m.group(2)) == cases and not fails, "\n".join(fails[:5]) + "\n" + tail


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_parity_against_the_oracle(seed):
    run_script("fuzz_parity.py", 60, seed)


@pytest.mark.parametrize("seed", [21, 22])
def test_randomised_decomposed_runs_follow_one_gpu(seed):
    run_script("fuzz_decomp.py", 16, seed)


def test_randomised_decomposed_runs_with_the_fused_drift_pass_on_small_boxes():
    """The fused bonded + kick + drift pass - on a decomposed handle it also packs the halo and adds the returned ghost forces -
    is a large-system arrangement (>= 2048 tiles per rank); MDX_WPT8_BELOW=32 takes the small boxes of the fuzz through it."""
    run_script("fuzz_decomp.py", 24, 23, {"MDX_WPT8_BELOW": "32"})


def test_randomised_decomposed_runs_on_twelve_and_sixteen_ranks_with_the_fold():
    """More peers than the seven of a 2 x 2 x 2 grid: the folded drift pass publishes its stale word to EVERY peer
    (FusedArgs::flag_rows holds one row per rank) and a partition that sends an atom to more than seven peers leaves the fold off
    (MdxDecomp::rows_fit); MDX_DD_SPEC_CHECK=1 holds every speculative drift probe against the synchronous one."""
    run_script("fuzz_decomp.py", 6, 24, {"MDX_WPT8_BELOW": "32", "FUZZ_WORLDS": "12,16", "MDX_DD_SPEC_CHECK": "1",
                                         "FUZZ_FORCE": "MDX_HALO_FOLD=1,MDX_HALF_SHELL=1,MDX_FUSE_BONDED_INTEGRATE_DD=1"})

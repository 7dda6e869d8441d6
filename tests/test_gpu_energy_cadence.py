"""Energies at a cadence (`mdx_set_energy_cadence`): the reference reads energies at the ratio of its snapshot handlers
(/root/reference src/md/mod.rs:121-122, src/properties/water_sol.rs:185-189), so the step loop evaluates them with the forces of
those steps and `mdx_energy` returns that evaluation.  What is checked: the held evaluation equals a fresh one on the same state,
it is really the one returned (counters), it is dropped when the state changes, and snapshots / the barostat use it too."""
import numpy as np
import pytest

from molchanica_amd import MdConfig, systems

pytestmark = pytest.mark.gpu

FIELDS = ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14", "kinetic", "potential", "temperature")


TERMS = ("bond", "angle", "dihedral", "lj", "coulomb", "lj14", "coulomb14")


def _close(a, b, rel=2e-6, floor=1e-3):
    scale = sum(abs(a[k]) for k in TERMS)       # (the potential is a sum of cancelling terms: fp32 rounding scales with them)
    for k in FIELDS:
        ref = scale if k == "potential" else max(abs(a[k]), abs(b[k]))
        assert abs(a[k] - b[k]) <= rel * ref + floor, (k, a[k], b[k])


def _state(system, cfg):
    from molchanica_amd.md_state import MdState
    return MdState(system, cfg)


@pytest.mark.parametrize("variant", [0, 2])
def test_held_evaluation_equals_a_fresh_one(variant):
    """Flexible water from the hot lattice, cadence 7: over 140 steps the cadence steps fall on chunk ends, behind list
    rebuilds and on pruning steps of the dual list.  At every cadence step the held energies are compared with an evaluation
    made afresh on the same handle (setting the cadence again drops the held one)."""
    s = systems.water_box(12, seed=3)
    with _state(s, MdConfig(skin=2.0, nb_variant=variant)) as md:
        md.initialize_velocities(300.0, seed=11)
        md.set_energy_cadence(7)
        served = 0
        for k in range(20):
            md.step(0.0005, None, 7)
            st0 = md.stats()
            held = md.energy()
            st1 = md.stats()
            assert st1["energies_from_step_loop"] == st0["energies_from_step_loop"] + 1     # no evaluation of its own
            assert st1["energy_evaluations"] == st0["energy_evaluations"]
            md.set_energy_cadence(7)
            fresh = md.energy()
            assert md.stats()["energy_evaluations"] == st1["energy_evaluations"] + 1
            _close(held, fresh)
            served += 1
        assert md.stats()["rebuild_count"] >= 3
        assert served == 20
        # off the cadence the call evaluates by itself, as before
        md.step(0.0005, None, 3)
        st0 = md.stats()
        md.energy()
        assert md.stats()["energy_evaluations"] == st0["energy_evaluations"] + 1


def test_same_trajectory_with_and_without_the_cadence():
    """The energy flavour of the force call computes the same forces (deterministic full-list kernel: the same bits), so a run
    told the cadence follows the run that is not."""
    s = systems.water_box(10, seed=5)
    cfg = MdConfig(skin=2.0, nb_variant=2, coulomb_mode=1)
    out = []
    for cadence in (0, 10):
        with _state(s, cfg) as md:
            md.initialize_velocities(300.0, seed=2)
            md.set_energy_cadence(cadence)
            es = []
            for k in range(6):
                md.step(0.0005, None, 10)
                es.append(md.energy())
            out.append((md.positions(), es, md.stats()))
    (pa, ea, sa), (pb, eb, sb) = out
    assert sb["energies_from_step_loop"] == 6 and sa["energies_from_step_loop"] == 0
    # six either way - inside the step loop, or by mdx_energy - plus the launches the device gated off behind a stale list
    # (the counter counts enqueued force calls of the energy flavour; the chunk's remainder then ends with another)
    assert sa["energy_evaluations"] == 6 and 6 <= sb["energy_evaluations"] <= 6 + sb["rebuild_count"]
    box = np.asarray(s.box_hi, np.float64) - np.asarray(s.box_lo, np.float64)
    d = pa.astype(np.float64) - pb
    d -= np.rint(d / box) * box
    assert np.abs(d).max() < 1e-4
    for x, y in zip(ea, eb):
        _close(x, y, rel=1e-5)


def test_held_evaluation_is_dropped_when_the_state_changes():
    s = systems.water_box(8, seed=9)
    with _state(s, MdConfig(skin=2.0)) as md:
        md.initialize_velocities(300.0, seed=4)
        md.set_energy_cadence(5)
        md.step(0.0005, None, 5)
        e0 = md.energy()
        v = md.velocities()
        md.set_velocities(0.5 * v)
        st0 = md.stats()
        e1 = md.energy()
        assert md.stats()["energy_evaluations"] == st0["energy_evaluations"] + 1
        assert abs(e1["kinetic"] - 0.25 * e0["kinetic"]) < 1e-4 * e0["kinetic"]
        assert abs(e1["potential"] - e0["potential"]) < 2e-6 * sum(abs(e0[k]) for k in TERMS) + 1e-3


def test_snapshots_and_the_barostat_read_the_held_evaluation():
    """Snapshots every 10 steps under a Berendsen thermostat every 5: a snapshot's energies are the step loop's evaluation, with
    the kinetic energy taken behind the thermostat's rescaling - the same numbers a caller gets who reads mdx_energy from
    outside at those steps of a run that was told nothing.  With a barostat every 10 steps the pressure comes from the step
    loop's evaluation and the box follows the same path."""
    s = systems.water_box(10, seed=6)
    cfg = MdConfig(skin=2.0, nb_variant=2, coulomb_mode=1)

    def run(snapshots, barostat):
        with _state(s, cfg) as md:
            md.initialize_velocities(300.0, seed=8)
            md.set_thermostat(1, 300.0, 0.1, 5, seed=1)
            if barostat:
                md.set_barostat(1, 1.0, 1.0, 4.6e-5, 10)
            if snapshots:
                md.set_snapshot_cadence(10)
            reads = []
            for k in range(4):
                md.step(0.0005, None, 10)
                if not snapshots:
                    reads.append(md.energy())
            if snapshots:
                reads = [sn["energy_data"] for sn in md.snapshots]
            return reads, md.stats(), md.cell()

    ra, sa, _ = run(True, False)
    rb, sb, _ = run(False, False)
    assert len(ra) == 4 and sa["energies_from_step_loop"] == 4 and sb["energies_from_step_loop"] == 0
    assert sb["energy_evaluations"] <= sa["energy_evaluations"] <= sb["energy_evaluations"] + sa["rebuild_count"]
    for x, y in zip(ra, rb):
        _close(x, y, rel=1e-5)
    rc, sc, cell_c = run(True, True)
    rd, sd, cell_d = run(False, True)
    assert sc["energies_from_step_loop"] >= 4 and sd["energies_from_step_loop"] >= 4       # the barostat's reads
    assert np.abs(np.asarray(cell_c[1]) - np.asarray(cell_d[1])).max() < 1e-4
    for x, y in zip(rc, rd):
        _close(x, y, rel=1e-5)

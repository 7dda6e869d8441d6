"""Multi-rank path on CPU: partition invariants, and 2-/4-rank `gloo` runs of tests/decomp_spec.py's DecomposedMd (with the
numpy engine double) that must reproduce the 1-rank trajectory.  Nothing here touches a GPU.

What this covers: the world-2 / world-4 rendezvous over `gloo`, the partition RULES (owners, halo membership, image shifts -
the same `Partition` class tests/test_gpu_partition_spec.py holds the device kernels of mdx_decomp.hip against), the
repartition and stale-flag protocol of the Python driver.  What it does not: the product's own step loop, transports and
half-shell force return, which live below the C ABI and need a GPU (tests/test_gpu_comm.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from molchanica_amd import MdConfig, MdSystem
from tests.decomp_spec import DecomposedMd, DistComm, Partition, ThreadComm, process_grid


def charged_fluid(n=360, box=30.0, seed=0):
    """Bond-free LJ + charge fluid on a jittered lattice (the engine double has no bonded terms)."""
    rng = np.random.default_rng(seed)
    m = int(np.ceil(n ** (1 / 3)))
    g = (np.arange(m) + 0.5) * box / m
    sites = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)[:n]
    pos = sites + rng.normal(scale=0.25, size=sites.shape)
    q = rng.choice([-0.4, 0.4], size=n)
    q -= q.mean()
    vel = rng.normal(scale=3.0, size=(n, 3))
    return MdSystem(pos=pos, vel=vel, mass=np.full(n, 16.0), charge=q, lj_type=np.zeros(n, int), lj_sigma=[3.0],
                    lj_eps=[0.15], periodic=True, box_lo=(0, 0, 0), box_hi=(box, box, box)).normalise()


CFG = MdConfig(lj_cutoff=6.0, coulomb_cutoff=6.0, skin=1.0, chunk_steps=5)


def test_process_grids():
    assert process_grid(1) == (1, 1, 1) and process_grid(2) == (2, 1, 1)
    assert process_grid(4) == (2, 2, 1) and process_grid(8) == (2, 2, 2)
    for w in (3, 6, 12):
        g = process_grid(w)
        assert g[0] * g[1] * g[2] == w


@pytest.mark.parametrize("world", [2, 4, 8])
def test_partition_invariants(world):
    s = charged_fluid(n=1500, box=40.0, seed=3)
    halo = 7.0
    part = Partition(s.box_lo, s.box_hi, world, halo)
    pos = part.wrap(torch.as_tensor(s.pos) + 55.0)        # exercise wrapping
    owner = part.owner(pos)
    assert int(owner.min()) >= 0 and int(owner.max()) < world
    assert torch.bincount(owner, minlength=world).sum() == s.n_atoms   # every atom owned exactly once
    L = np.array(s.box_hi, dtype=np.float64)
    p = pos.numpy().astype(np.float64)
    for r in range(world):
        mask, shift = part.local_mask_and_shift(r, pos)
        blo, bhi = part.brick(r)
        owned = (owner == r).numpy()
        assert mask.numpy()[owned].all()
        # reference: minimum-image distance of each atom to the brick, decomposed dimensions only
        inside = np.ones(s.n_atoms, bool)
        for d in range(3):
            if part.grid[d] == 1:
                continue
            c = 0.5 * (blo[d] + bhi[d])
            dx = p[:, d] - c
            dx -= np.rint(dx / L[d]) * L[d]
            inside &= np.abs(dx) < 0.5 * (bhi[d] - blo[d]) + halo - 1e-4
        m = mask.numpy()
        assert (m | ~inside).all(), "an atom within the halo is missing"
        loose = np.ones(s.n_atoms, bool)
        for d in range(3):
            if part.grid[d] == 1:
                continue
            c = 0.5 * (blo[d] + bhi[d])
            dx = p[:, d] - c
            dx -= np.rint(dx / L[d]) * L[d]
            loose &= np.abs(dx) <= 0.5 * (bhi[d] - blo[d]) + halo + 1e-4
        assert (loose | ~m).all(), "an atom outside the halo was selected"
        # shifted coordinates land inside the padded brick
        ps = (pos + shift).numpy()[m]
        for d in range(3):
            if part.grid[d] > 1:
                assert (ps[:, d] >= blo[d] - halo - 1e-3).all() and (ps[:, d] < bhi[d] + halo + 1e-3).all()
    # send/receive lists agree: what q receives from r is what r sends to q
    for r in range(world):
        mr, _ = part.local_mask_and_shift(r, pos)
        for q in range(world):
            if q == r:
                continue
            mq, _ = part.local_mask_and_shift(q, pos)
            send_r_to_q = torch.nonzero(mq & (owner == r)).flatten()
            recv_q_from_r = torch.nonzero(mq & (owner == r)).flatten()
            assert torch.equal(send_r_to_q, recv_q_from_r)


def test_partition_rejects_thin_boxes():
    with pytest.raises(ValueError, match="two images"):
        Partition((0, 0, 0), (30, 30, 30), 2, 9.0)


def _reference_trajectory(n_steps, box=30.0):
    from tests.engine_double import NumpyEngine
    s = charged_fluid(n=int(360 * (box / 30.0) ** 3), box=box)
    md = DecomposedMd(s, CFG, rank=0, world=1, engine=NumpyEngine(s, CFG))
    md.step(0.002, n_steps)
    return md.positions(), md.velocities(), md.repartitions


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _worker(rank, world, port, n_steps, out_dir, margin, box):
    from tests.engine_double import NumpyEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        s = charged_fluid(n=int(360 * (box / 30.0) ** 3), box=box)
        eng = NumpyEngine(s, CFG)
        md = DecomposedMd(s, CFG, rank=rank, world=world, engine=eng, comm=DistComm(rank, world), halo_margin=margin)
        assert md.n_owned + md.stats()["n_ghost"] == eng.n_local
        md.step(0.002, n_steps)
        pos, vel = md.positions(), md.velocities()
        # every rank reconstructs the same global state
        t = torch.from_numpy(pos.copy())
        dist.broadcast(t, 0)
        assert np.array_equal(t.numpy(), pos)
        if rank == 0:
            np.savez(os.path.join(out_dir, "out.npz"), pos=pos, vel=vel, rep=md.repartitions,
                     owned=md.n_owned, local=eng.n_local, local_rebuilds=md.local_rebuilds_total)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,margin", [(2, 0.0), (4, 0.0), (2, 1.4)])
def test_gloo_ranks_reproduce_single_domain(world, margin, tmp_path):
    """margin 0: every stale list repartitions; margin 1.4 Å: local rebuilds while the measured drift since the
    last repartition stays below 0.7 Å, then a repartition."""
    n_steps = 23 if margin == 0.0 else 40
    box = 30.0 if margin == 0.0 else 36.0
    ref_pos, ref_vel, ref_rep = _reference_trajectory(n_steps, box)
    mp.spawn(_worker, args=(world, _free_port(), n_steps, str(tmp_path), margin, box), nprocs=world, join=True)
    out = np.load(tmp_path / "out.npz")
    L = box
    d = out["pos"] - ref_pos
    d -= np.round(d / L) * L
    assert np.abs(d).max() < 2e-4, np.abs(d).max()          # f32 hand-over at repartition, fp64 inside
    assert np.abs(out["vel"] - ref_vel).max() < 2e-3
    assert int(out["rep"]) >= 2, "no repartition happened: the test would not cover migration"
    assert (int(out["local_rebuilds"]) == 0) if margin == 0.0 else (int(out["local_rebuilds"]) >= 1)
    assert int(out["owned"]) < ref_pos.shape[0] and int(out["local"]) > int(out["owned"])


def test_thread_comm_matches_single_domain():
    """The in-process communicator (used on the single-GPU box) drives the same protocol."""
    import threading
    from tests.engine_double import NumpyEngine
    n_steps, world = 17, 2
    ref_pos, _, _ = _reference_trajectory(n_steps)
    shared = ThreadComm.Shared(world)
    res, errs = {}, []

    def run(rank):
        try:
            s = charged_fluid()
            md = DecomposedMd(s, CFG, rank=rank, world=world, engine=NumpyEngine(s, CFG), comm=ThreadComm(rank, shared),
                              halo_margin=0.0)
            md.step(0.002, n_steps)
            res[rank] = md.positions()
        except Exception as e:   # pragma: no cover
            errs.append(e)
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    d = res[0] - ref_pos
    d -= np.round(d / 30.0) * 30.0
    assert np.abs(d).max() < 2e-4 and np.array_equal(res[0], res[1])

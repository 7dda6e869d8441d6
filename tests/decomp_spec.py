"""EXECUTABLE SPECIFICATION of the spatial decomposition (tests only).

The product's decomposition lives below the C ABI (molchanica_amd/csrc/mdx_decomp.hip, mdx_comm.hip: `mdx_comm_init`); this
file is its statement in plain torch, kept under tests/ for two jobs:
  * `Partition` - bricks, owners, halo membership, image shifts, the half-shell rule and the bonded-partner rule - is what
    tests/test_gpu_partition_spec.py holds the device kernels of mdx_decomp.hip against, atom by atom, at 2 / 4 / 8 ranks;
  * `DecomposedMd` - the round-1 host-driven step loop over the ABI's building blocks (mdx_set_local_atoms, mdx_chunk_*,
    mdx_pack/unpack_positions: still public, for hosts that drive a decomposition themselves) - runs under `gloo` with a numpy
    engine double in tests/test_decomp_gloo.py (world 2 and 4 on CPU: rendezvous, repartition, halo lists, stale-flag
    protocol of THIS driver, full-shell halo) and on the HIP engine in tests/test_gpu_decomp.py.
What the gloo test does NOT cover: the C++ transports, the half-shell force return and the device partition kernels - those
are tests/test_gpu_comm.py and tests/test_gpu_partition_spec.py.

The reference is single-device (`CudaContext::new(0)`, /root/reference src/util.rs:1086; no
NCCL/MPI anywhere in the tree), so this is new capability layered on the same C ABI:

  * the box is cut into P = px*py*pz bricks (2 -> 2x1x1, 4 -> 2x2x1, 8 -> 2x2x2; with periodic
    wrap every rank of a 2x2x2 grid has exactly 7 distinct peers = the 7 xGMI links of an MI355X);
  * a rank integrates the atoms it OWNS (those inside its brick at the last repartition) and
    keeps GHOST copies of every other atom within `halo = cutoff + skin` of the brick, already
    shifted into its own frame - a decomposed dimension is therefore not periodic locally, a
    dimension that is not cut stays periodic inside the engine;
  * (DecomposedMd) every step: drift -> 1-word all-reduce(max) of the rebuild flag -> ONE batched group of
    point-to-point sends/receives of ghost positions, packed and unpacked by index-list kernels on the
    engine's stream -> forces, full shell: no force message travels back;
  * when any rank's list goes stale, all ranks repartition from the same gathered data with the same arithmetic - the
    lists agree by construction, sorted by global atom id, and no index list is ever exchanged.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.distributed as dist

from molchanica_amd._abi import MdConfig, MdSystem

GRIDS = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}


def process_grid(world: int) -> tuple[int, int, int]:
    if world in GRIDS:
        return GRIDS[world]
    # generic: peel prime factors onto the currently shortest grid axis
    g = [1, 1, 1]
    n, f = world, 2
    while n > 1:
        while n % f == 0:
            g[int(np.argmin(g))] *= f
            n //= f
        f += 1
    return tuple(sorted(g, reverse=True))


class Partition:
    """Bricks, owners, halos and image shifts.  Pure torch; identical results on every rank."""

    def __init__(self, box_lo, box_hi, world: int, halo: float):
        self.world = world
        self.grid = process_grid(world)
        self.lo = torch.tensor(box_lo, dtype=torch.float32)
        self.hi = torch.tensor(box_hi, dtype=torch.float32)
        self.len = self.hi - self.lo
        self.halo = float(halo)
        for d in range(3):
            if self.grid[d] > 1:
                brick = float(self.len[d]) / self.grid[d]
                if brick + 2 * self.halo > float(self.len[d]) + 1e-3:
                    raise ValueError(f"dimension {d}: brick {brick:.1f} + 2*halo {self.halo:.1f} exceeds the box "
                                     f"{float(self.len[d]):.1f}: an atom would be needed under two images")

    def coords(self, rank: int) -> tuple[int, int, int]:
        px, py, pz = self.grid
        return rank // (py * pz), (rank // pz) % py, rank % pz

    def brick(self, rank: int):
        c = self.coords(rank)
        lo = [float(self.lo[d]) + float(self.len[d]) * c[d] / self.grid[d] for d in range(3)]
        hi = [float(self.lo[d]) + float(self.len[d]) * (c[d] + 1) / self.grid[d] for d in range(3)]
        return lo, hi

    def periodic_mask(self) -> int:
        """MDX_PERIODIC_DIMS of the LOCAL region: a dimension that is not cut stays periodic."""
        g = self.grid
        return 0x10 | (1 if g[0] == 1 else 0) | (2 if g[1] == 1 else 0) | (4 if g[2] == 1 else 0)

    def wrap(self, pos: torch.Tensor) -> torch.Tensor:
        lo, ln = self.lo.to(pos.device), self.len.to(pos.device)
        p = pos - torch.floor((pos - lo) / ln) * ln
        p = torch.where(p < lo, p + ln, p)
        p = torch.where(p >= lo + ln, p - ln, p)
        return p

    def owner(self, pos_wrapped: torch.Tensor) -> torch.Tensor:
        dev = pos_wrapped.device
        lo, ln = self.lo.to(dev), self.len.to(dev)
        g = torch.tensor(self.grid, device=dev, dtype=torch.float32)
        c = torch.floor((pos_wrapped - lo) / ln * g).to(torch.int64)
        c = torch.minimum(torch.clamp(c, min=0), (g.to(torch.int64) - 1))
        return (c[:, 0] * self.grid[1] + c[:, 1]) * self.grid[2] + c[:, 2]

    def local_mask_and_shift(self, rank: int, pos_wrapped: torch.Tensor):
        """-> (mask [N] bool: atom is simulated by `rank` (owned or ghost), shift [N,3] image shift
        that moves the atom into rank's frame)."""
        dev = pos_wrapped.device
        # brick faces and the widened interval in fp32, operation by operation as dd_in_halo (mdx_decomp.hip) evaluates them
        f32 = np.float32
        c = self.coords(rank)
        mask = torch.ones(pos_wrapped.shape[0], dtype=torch.bool, device=dev)
        shift = torch.zeros_like(pos_wrapped)
        for d in range(3):
            if self.grid[d] == 1:
                continue
            x = pos_wrapped[:, d]
            L = float(self.len[d])
            lo32, len32 = f32(self.lo[d]), f32(self.len[d])
            blo = f32(lo32 + f32(f32(len32 * f32(c[d])) / f32(self.grid[d])))
            bhi = f32(lo32 + f32(f32(len32 * f32(c[d] + 1)) / f32(self.grid[d])))
            lo_h, hi_h = float(f32(blo - f32(self.halo))), float(f32(bhi + f32(self.halo)))
            any_k = torch.zeros_like(mask)
            sh = torch.zeros_like(x)
            for k in (-1.0, 0.0, 1.0):
                xs = x + k * L
                ink = (xs >= lo_h) & (xs < hi_h)
                sh = torch.where(ink & ~any_k, torch.full_like(x, k * L), sh)
                any_k |= ink
            mask &= any_k
            shift[:, d] = sh
        return mask, shift

    # ---- the rules mdx_decomp.hip adds (dd_classify_kernel), restated ---------------------------------------------------
    def image_codes(self, rank: int, pos_wrapped: torch.Tensor):
        """-> (here [N] bool, k [N,3] int: the image (-1, 0, +1 per cut dimension, the FIRST that fits) under which the atom
        lies inside `rank`'s brick widened by the halo)."""
        mask, shift = self.local_mask_and_shift(rank, pos_wrapped)
        k = torch.zeros(pos_wrapped.shape, dtype=torch.int64)
        for d in range(3):
            if self.grid[d] > 1:
                k[:, d] = torch.round(shift[:, d] / float(self.len[d])).to(torch.int64)
        return mask, k

    def classify(self, rank: int, pos_wrapped: torch.Tensor, owner: torch.Tensor, half_shell: bool, role_partners=None):
        """Class of every atom on `rank`: 0 not here, 1 owned, 2 ghost, 3 ghost kept only as the bonded partner of an owned atom.
        Half shell: a ghost is kept only if its OWNER's brick - in the frame the atom lives in: its image here minus the image its
        owner holds it under - lies in an upper direction (first non-zero component of  c_owner + (k_here - k_home) grid - c_rank
        positive).  role_partners: list of (atom, partner) pairs of the bonded terms (both directions)."""
        n = pos_wrapped.shape[0]
        here, k_here = self.image_codes(rank, pos_wrapped)
        cls = torch.where(owner == rank, 1, torch.where(here, 2, 0)).to(torch.int64)
        if half_shell:
            k_home = torch.zeros_like(k_here)
            for q in range(self.world):
                sel = owner == q
                if sel.any():
                    _, kq = self.image_codes(q, pos_wrapped)
                    k_home[sel] = kq[sel]
            cr = torch.tensor(self.coords(rank))
            g = torch.tensor(self.grid)
            co = torch.stack([owner // (self.grid[1] * self.grid[2]), (owner // self.grid[2]) % self.grid[1], owner % self.grid[2]], 1)
            rel = co + (k_here - k_home) * g - cr
            rel[:, [d for d in range(3) if self.grid[d] == 1]] = 0
            upper = torch.zeros(n, dtype=torch.bool); decided = torch.zeros(n, dtype=torch.bool)
            for d in range(3):
                nz = (rel[:, d] != 0) & ~decided
                upper |= nz & (rel[:, d] > 0)
                decided |= nz
            cls = torch.where((cls == 2) & ~upper, 0, cls)
            if role_partners is not None and len(role_partners):
                a, b = role_partners[:, 0], role_partners[:, 1]
                need = (owner[b] == rank) & (owner[a] != rank)          # atom a shares a term with an atom this rank owns
                wanted = torch.zeros(n, dtype=torch.bool); wanted[a[need]] = True
                cls = torch.where(wanted & (cls == 0) & here, 3, cls)
        return cls, k_here

    def send_mask(self, rank: int, pos_wrapped: torch.Tensor, owner: torch.Tensor, half_shell: bool, role_partners=None):
        """For the atoms `rank` owns: bit q <=> rank q keeps a copy (the mirror image of classify on q)."""
        m = torch.zeros(pos_wrapped.shape[0], dtype=torch.int64)
        for q in range(self.world):
            if q == rank:
                continue
            cls_q, _ = self.classify(q, pos_wrapped, owner, half_shell, role_partners)
            m |= ((cls_q >= 2) & (owner == rank)).to(torch.int64) << q
        return m

    def local_bounds(self, rank: int, pad: float = 1.0):
        blo, bhi = self.brick(rank)
        lo = [blo[d] - self.halo - pad if self.grid[d] > 1 else float(self.lo[d]) for d in range(3)]
        hi = [bhi[d] + self.halo + pad if self.grid[d] > 1 else float(self.hi[d]) for d in range(3)]
        return lo, hi


class DistComm:
    """torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests)."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world = rank, world

    def all_reduce(self, t: torch.Tensor, op: str):
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)

    def exchange(self, sends, recvs):
        """sends: [(peer, tensor)], recvs: [(peer, tensor)] -> one batched group of point-to-point ops
        (ncclGroupStart / ncclSend / ncclRecv ... / ncclGroupEnd under the nccl backend)."""
        self.run(self.prepare(sends, recvs))

    def prepare(self, sends, recvs):
        """The op list of a halo exchange is the same every step until the next repartition (same peers, same
        slices of the same buffers): build the P2POp objects once."""
        return [dist.P2POp(dist.isend, t, q) for q, t in sends] + [dist.P2POp(dist.irecv, t, q) for q, t in recvs]

    def prepare_halo(self, send_buf, recv_buf, send_segs, recv_segs):
        """Halo exchange over ONE send and ONE receive buffer whose per-peer segments are stored in rank order
        (send_segs / recv_segs: [(peer, row0, row1)]).  Default: a single `all_to_all_single` with uneven
        splits - under the nccl backend one ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd, i.e.
        the same wire traffic as the batched point-to-point list, at a fraction of its per-step Python cost
        (one collective call instead of 2 x peers work objects).  MDX_HALO_P2P=1 selects the P2P list."""
        import os
        if os.environ.get("MDX_HALO_P2P", "0") == "1":
            return self.prepare([(q, send_buf[a:b]) for q, a, b in send_segs], [(q, recv_buf[a:b]) for q, a, b in recv_segs])
        ins, outs = [0] * self.world, [0] * self.world
        for q, a, b in send_segs:
            ins[q] = b - a
        for q, a, b in recv_segs:
            outs[q] = b - a
        return ("a2a", send_buf, recv_buf, ins, outs)

    def run(self, ops):
        if isinstance(ops, tuple) and ops and ops[0] == "a2a":
            _, send_buf, recv_buf, ins, outs = ops
            dist.all_to_all_single(recv_buf, send_buf, output_split_sizes=outs, input_split_sizes=ins)
            return
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()      # nccl: the current stream waits for the transfer; the host does not block


class ThreadComm:
    """In-process stand-in for the communicator: `world` ranks run as threads of ONE process (each
    with its own engine handle and stream, possibly all on the same GPU) and meet at barriers.
    Used to exercise the complete multi-rank path — ghosts, image shifts, non-periodic local
    regions, halo traffic, repartition — on a single-GPU box, where RCCL itself cannot be run."""

    class Shared:
        def __init__(self, world: int):
            import threading
            self.world = world
            self.barrier = threading.Barrier(world)
            self.slots = [None] * world
            self.mail = {}

    def __init__(self, rank: int, shared: "ThreadComm.Shared"):
        self.rank, self.world, self.sh = rank, shared.world, shared

    @staticmethod
    def _sync(t: torch.Tensor):
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()

    def all_reduce(self, t: torch.Tensor, op: str):
        sh = self.sh
        self._sync(t)
        sh.slots[self.rank] = t
        sh.barrier.wait()
        stack = torch.stack([x.to(t.device) for x in sh.slots])
        red = stack.max(0).values if op == "max" else stack.sum(0)
        self._sync(t)
        sh.barrier.wait()          # everybody has read everybody's input
        t.copy_(red)
        self._sync(t)
        sh.barrier.wait()

    def prepare(self, sends, recvs):
        return (sends, recvs)

    def prepare_halo(self, send_buf, recv_buf, send_segs, recv_segs):
        return ([(q, send_buf[a:b]) for q, a, b in send_segs], [(q, recv_buf[a:b]) for q, a, b in recv_segs])

    def run(self, prepared):
        self.exchange(*prepared)

    def exchange(self, sends, recvs):
        sh = self.sh
        for q, t in sends:
            self._sync(t)
            sh.mail[(self.rank, q)] = t
        sh.barrier.wait()
        for q, t in recvs:
            t.copy_(sh.mail[(q, self.rank)])
            self._sync(t)
        sh.barrier.wait()
        for q, t in sends:
            sh.mail.pop((self.rank, q), None)
        sh.barrier.wait()


class HipEngine:
    """Adapter: torch tensors -> device pointers of the C ABI (molchanica_amd.md_state.MdState)."""

    def __init__(self, system: MdSystem, cfg: MdConfig, device: int):
        from molchanica_amd.md_state import MdState
        self.md = MdState(system, cfg, device)
        self.device = torch.device("cuda", device)
        self.stream = torch.cuda.ExternalStream(self.md.stream_ptr(), device=self.device)
        self._flags = None
        self._keep = []

    def set_local_atoms(self, gid, ghost, pos4, vel4, lo, hi, periodic_mask):
        self._keep = [gid, ghost, pos4, vel4]
        self.n_local = int(gid.numel())
        self.md.set_local_atoms(self.n_local, gid.data_ptr(), ghost.data_ptr(), pos4.data_ptr(), vel4.data_ptr(),
                                lo, hi, periodic_mask)

    def local_state(self):
        pos4 = torch.empty((self.n_local, 4), dtype=torch.float32, device=self.device)
        vel4 = torch.empty_like(pos4)
        self.md.local_state(pos4.data_ptr(), vel4.data_ptr())
        return pos4, vel4

    def flag_tensor(self) -> torch.Tensor:
        if self._flags is None:
            ptr = self.md.flag_words_ptr()

            class _View:
                __cuda_array_interface__ = {"shape": (66,), "typestr": "<i4", "data": (ptr, False), "version": 2}

            self._flags = torch.as_tensor(_View(), device=self.device)
        return self._flags

    def stale_threshold(self) -> int:
        return self.md.stale_threshold()

    def chunk_begin(self): self.md.chunk_begin()
    def chunk_integrate(self, mode, dt, s): self.md.chunk_integrate(mode, dt, s)
    def chunk_forces(self, s): self.md.chunk_forces(s)
    def chunk_end(self, n): return self.md.chunk_end(n)
    def add_steps(self, n): self.md.add_steps(n)

    def pack(self, gid, out4, flag_word=-1):
        self.md.pack_positions(gid.data_ptr(), int(gid.numel()), out4.data_ptr(), flag_word)

    def unpack(self, gid, in4, shift4, flag_word=-1):
        self.md.unpack_positions(gid.data_ptr(), int(gid.numel()), in4.data_ptr(), shift4.data_ptr(), flag_word)

    def rebuild(self): self.md.rebuild_spatial_caches()
    def energy(self): return self.md.energy()
    def stats(self): return self.md.stats()
    def profile(self, on): self.md.profile(on)


class DecomposedMd:
    """`MdState`-like stepping of ONE box decomposed over `world` ranks (one process per GPU)."""

    def __init__(self, system: MdSystem, cfg: MdConfig, rank: int, world: int, device: int = 0,
                 engine=None, halo_margin: float = 4.4, comm=None):
        if not system.periodic:
            raise ValueError("spatial decomposition needs a periodic box")
        self.system, self.cfg, self.rank, self.world = system.normalise(), cfg, rank, world
        self.n_atoms = system.n_atoms
        # Ghosts are kept out to r_list + margin.  Atoms may then drift margin/2 from where they
        # were at the last repartition before any rank can miss a neighbour, so that many local
        # list rebuilds (each triggered by skin/2 of drift, on the current owned+ghost set, no host
        # work, no ownership change) are allowed between two repartitions.
        r_list = max(cfg.lj_cutoff, cfg.coulomb_cutoff) + cfg.skin
        grid = process_grid(world)
        room = min([(float(system.box_hi[d]) - float(system.box_lo[d])) * (1.0 - 1.0 / grid[d]) / 2.0 - r_list - 0.01
                    for d in range(3) if grid[d] > 1] or [halo_margin])
        self.halo_margin = max(0.0, min(float(halo_margin), room))
        self.halo = r_list + self.halo_margin
        # How long is that?  Counting rebuilds (each "uses up" skin/2 + overshoot of the budget) is the worst
        # case and allowed ONE local rebuild per repartition; measuring is better: at every rebuild event the
        # largest displacement since the last repartition is reduced over the ranks (one small all-reduce), and
        # the local atom set is kept while it stays below margin/2.  At 300 K that is hundreds of steps, not ~50.
        self.local_rebuilds = 0
        self.local_rebuilds_total = 0
        self.part = Partition(system.box_lo, system.box_hi, world, self.halo)
        self.engine = engine if engine is not None else HipEngine(system, cfg, device)
        self.comm = comm if comm is not None else DistComm(rank, world)
        self.dev = self.engine.device
        self.chunk = max(1, min(int(cfg.chunk_steps), 64))
        self._flag_tmp = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self.step_count = 0
        self.repartitions = 0
        self.repartition_s = 0.0
        pos = torch.as_tensor(self.system.pos, dtype=torch.float32).to(self.dev)
        vel = torch.as_tensor(self.system.vel if self.system.vel is not None
                              else np.zeros_like(self.system.pos), dtype=torch.float32).to(self.dev)
        with self._stream():
            self._repartition_from(pos, vel)
            self.engine.chunk_begin()
            self.engine.chunk_forces(-1)

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        s = getattr(self.engine, "stream", None)
        return torch.cuda.stream(s) if s is not None else _NullCtx()

    def describe(self) -> str:
        g = self.part.grid
        return (f"spatial {g[0]}x{g[1]}x{g[2]} bricks, ghost halo {self.halo:.1f} A, RCCL send/recv per peer, "
                f"local list rebuilds while max drift < {0.5 * self.halo_margin:.1f} A, then repartition")

    def _repartition_from(self, pos_all: torch.Tensor, vel_all: torch.Tensor):
        """pos_all/vel_all: [N,3] global state, identical on every rank."""
        part, r = self.part, self.rank
        posw = part.wrap(pos_all)
        owner = part.owner(posw)
        mask, shift = part.local_mask_and_shift(r, posw)
        owned = owner == r
        assert bool((mask | ~owned).all()), "an owned atom fell outside its own halo region"
        self.owned_gid = torch.nonzero(owned, as_tuple=False).flatten()
        self.n_owned = int(self.owned_gid.numel())
        gid_local = torch.nonzero(mask, as_tuple=False).flatten()            # ascending global ids
        ghost = (~owned[gid_local]).to(torch.uint8)
        self.own_rows = owned[gid_local]
        pos_l = posw[gid_local] + shift[gid_local]
        n = gid_local.numel()
        pos4 = torch.zeros((n, 4), dtype=torch.float32, device=self.dev)
        vel4 = torch.zeros_like(pos4)
        pos4[:, :3] = pos_l
        vel4[:, :3] = vel_all[gid_local]
        lo, hi = part.local_bounds(r)
        self.gid_local = gid_local.to(torch.int32).contiguous()
        self.engine.set_local_atoms(self.gid_local, ghost.contiguous(), pos4.contiguous(), vel4.contiguous(),
                                    lo, hi, part.periodic_mask())
        self.pos_at_part = pos4[:, :3].clone()
        # halo lists, derived identically on every rank: what rank q needs from rank p.  All peers'
        # rows live in ONE send and ONE receive buffer (one pack and one unpack launch per step); a
        # peer's segment ends with a flag row (id -1) that carries the rebuild-flag word.
        INV = -1
        s_ids, r_ids, r_shift, self.send, self.recv = [], [], [], [], []   # (peer, row0, row1)
        s0 = r0 = 0
        own_idx = self.owned_gid                      # my atoms: candidates for every peer's halo
        pos_own = posw[own_idx]
        owner_loc = owner[gid_local]                  # owners of everything simulated here
        shift_loc = shift[gid_local]
        for q in range(self.world):
            if q == r:
                continue
            mq, _ = part.local_mask_and_shift(q, pos_own)                # O(N/world) per peer
            s_idx = own_idx[mq]                                          # mine, ghost on q (ascending ids)
            sel = owner_loc == q
            r_idx = gid_local[sel]                                       # q's, ghost here (ascending ids)
            if s_idx.numel():
                s_ids += [s_idx.to(torch.int32), torch.full((1,), INV, dtype=torch.int32, device=self.dev)]
                self.send.append((q, s0, s0 + s_idx.numel() + 1))
                s0 += s_idx.numel() + 1
            if r_idx.numel():
                r_ids += [r_idx.to(torch.int32), torch.full((1,), INV, dtype=torch.int32, device=self.dev)]
                sh = torch.zeros((r_idx.numel() + 1, 4), dtype=torch.float32, device=self.dev)
                sh[:-1, :3] = shift_loc[sel]
                r_shift.append(sh)
                self.recv.append((q, r0, r0 + r_idx.numel() + 1))
                r0 += r_idx.numel() + 1
        mk = lambda xs, shape, dt: (torch.cat(xs).contiguous() if xs else torch.zeros(shape, dtype=dt, device=self.dev))
        self.send_ids = mk(s_ids, (0,), torch.int32)
        self.recv_ids = mk(r_ids, (0,), torch.int32)
        self.recv_shift = mk(r_shift, (0, 4), torch.float32)
        self.send_buf = torch.zeros((s0, 4), dtype=torch.float32, device=self.dev)
        self.recv_buf = torch.zeros((r0, 4), dtype=torch.float32, device=self.dev)
        self._halo_ops = None          # rebuilt lazily: new buffers, new slices
        # the flag can ride on the halo only if every other rank is a peer in both directions
        # ... on EVERY rank: the per-step flag all-reduce is a collective, so all ranks must take the same branch.
        # One small all-reduce per repartition settles it (an empty or sparse brick may lack a peer segment).
        lacking = torch.tensor([0 if (len(self.send) == self.world - 1 and len(self.recv) == self.world - 1) else 1],
                               dtype=torch.int32, device=self.dev)
        if self.world > 1:
            self.comm.all_reduce(lacking, "max")
        self.flag_on_halo = int(lacking.item()) == 0
        self.repartitions += 1

    def _local_set_still_valid(self) -> bool:
        """May the owned + ghost set of the last repartition serve one more list rebuild?  Yes while no atom has
        moved further than margin/2 from where it was then (a stranger and an owned atom approaching each other
        can then not have closed the `margin` that separated the stranger from the halo)."""
        if self.world == 1:
            return True                       # nothing is left out of a one-rank box
        if self.halo_margin <= 0.0 or self.local_rebuilds >= 256:
            return False
        pos4, _ = self.engine.local_state()
        d = pos4[:, :3] - self.pos_at_part
        L = self.part.len.to(d.device)
        for k in range(3):                    # an uncut dimension stays periodic inside the engine: it may have wrapped
            if self.part.grid[k] == 1:
                d[:, k] -= torch.round(d[:, k] / L[k]) * L[k]
        d2 = (d * d).sum(1).max().reshape(1)
        self.comm.all_reduce(d2, "max")
        return float(d2.item()) ** 0.5 <= 0.5 * self.halo_margin - 0.05

    def _gather_global(self):
        """-> (pos [N,3], vel [N,3]) of the whole box on every rank: owned rows scattered into a zero
        array, one all-reduce(sum); every row is non-zero on exactly one rank, so the sum is exact."""
        pos4, vel4 = self.engine.local_state()
        state = torch.zeros((self.n_atoms, 6), dtype=torch.float32, device=self.dev)
        gl = self.gid_local.to(torch.int64)
        own_rows = self.own_rows
        state[gl[own_rows], :3] = pos4[own_rows, :3]
        state[gl[own_rows], 3:] = vel4[own_rows, :3]
        if self.world > 1:
            self.comm.all_reduce(state, "sum")
        return state[:, :3].contiguous(), state[:, 3:].contiguous()

    def _halo_exchange(self, flag_word: int):
        if self.world == 1:
            return
        fw = flag_word if self.flag_on_halo else -1
        if self.send_ids.numel():
            self.engine.pack(self.send_ids, self.send_buf, fw)
        if self._halo_ops is None:
            self._halo_ops = self.comm.prepare_halo(self.send_buf, self.recv_buf, self.send, self.recv)
        self.comm.run(self._halo_ops)
        if self.recv_ids.numel():
            self.engine.unpack(self.recv_ids, self.recv_buf, self.recv_shift, fw)

    # ------------------------------------------------------------------------------------------
    def step(self, dt: float, n_steps: int):
        eng = self.engine
        thr = eng.stale_threshold()
        remaining = int(n_steps)
        with self._stream():
            while remaining > 0:
                chunk = min(remaining, self.chunk)
                eng.chunk_begin()
                flags = eng.flag_tensor()
                for s in range(chunk):
                    eng.chunk_integrate(0 if s == 0 else 1, dt, s)
                    if self.world > 1 and not self.flag_on_halo:
                        # reduce through a torch-owned word: the flag array itself lives in the
                        # engine's allocation, which the NCCL process group does not know about
                        self._flag_tmp.copy_(flags[s + 1:s + 2])
                        self.comm.all_reduce(self._flag_tmp, "max")
                        flags[s + 1:s + 2].copy_(self._flag_tmp)
                    self._halo_exchange(s + 1)
                    eng.chunk_forces(s)
                eng.chunk_integrate(2, dt, chunk)
                words = eng.chunk_end(chunk + 1)
                done = chunk
                for s in range(chunk):
                    if int(words[s + 1]) > thr:
                        if np.uint32(words[s + 1]).view(np.float32) > 1.0e29:
                            raise FloatingPointError("non-finite or runaway coordinates during decomposed step")
                        # the drift of step s happened everywhere, its forces nowhere: repartition, finish it
                        if self._local_set_still_valid():      # same decision on every rank: reduced over all
                            self.local_rebuilds += 1; self.local_rebuilds_total += 1
                            eng.rebuild()
                        else:
                            import time as _t
                            t0 = _t.perf_counter()
                            pos, vel = self._gather_global()
                            self._repartition_from(pos, vel)
                            if self.dev.type == "cuda":
                                torch.cuda.current_stream(self.dev).synchronize()
                            self.repartition_s += _t.perf_counter() - t0
                            self.local_rebuilds = 0
                        eng.chunk_begin()
                        eng.chunk_forces(-1)
                        eng.chunk_integrate(2, dt, 0)
                        done = s + 1
                        break
                remaining -= done
                self.step_count += done
                eng.add_steps(done)

    def positions(self) -> np.ndarray:
        with self._stream():
            pos, _ = self._gather_global()
        return self.part.wrap(pos).cpu().numpy()

    def velocities(self) -> np.ndarray:
        with self._stream():
            _, vel = self._gather_global()
        return vel.cpu().numpy()

    def energy(self) -> dict:
        e = self.engine.energy()
        keys = ["kinetic", "lj", "coulomb", "lj14", "coulomb14", "bond", "angle", "dihedral", "virial"]
        t = torch.tensor([e[k] for k in keys], dtype=torch.float64, device=self.dev)
        if self.world > 1:
            self.comm.all_reduce(t, "sum")
        out = dict(zip(keys, (float(v) for v in t.cpu())))
        out["potential_bonded"] = out["bond"] + out["angle"] + out["dihedral"]
        out["potential_nonbonded"] = out["lj"] + out["coulomb"] + out["lj14"] + out["coulomb14"]
        out["potential"] = out["potential_bonded"] + out["potential_nonbonded"]
        vol = float(torch.prod(self.part.len).item()) if self.system.periodic else 0.0
        out["volume"] = vol
        out["pressure"] = (2.0 * out["kinetic"] + out["virial"]) / (3.0 * vol) * 69476.95 if vol > 0 else 0.0
        return out

    def stats(self) -> dict:
        st = self.engine.stats()
        st["repartitions"] = self.repartitions
        st["repartition_ms_sum"] = 1e3 * self.repartition_s
        st["n_owned"] = self.n_owned
        st["n_ghost"] = int(self.gid_local.numel()) - self.n_owned
        return st

    def profile(self, on: bool = True):
        self.engine.profile(on)


class _NullCtx:
    def __enter__(self): return self
    def __exit__(self, *a): return False

"""CPU test double of the engine interface tests.decomp_spec.DecomposedMd drives.

Pure numpy, fp64: non-bonded LJ + shifted-cutoff Coulomb by brute force over the LOCAL atom set
(minimum image only in the dimensions that are periodic locally), velocity-Verlet kick/drift and
the same flag-word protocol as the HIP library.  It exists so that the decomposition logic —
owners, ghosts, image shifts, halo traffic, repartition — can be verified with world_size > 1 on a
CPU-only host under `gloo`; it is not a product path."""
from __future__ import annotations

import numpy as np
import torch

ACC = 418.4
KE = 332.0637


class NumpyEngine:
    device = torch.device("cpu")
    stream = None

    def __init__(self, system, cfg):
        s = system
        self.N = s.n_atoms
        self.q = s.charge.astype(np.float64)
        self.sig = s.lj_sigma[s.lj_type].astype(np.float64)
        self.eps = s.lj_eps[s.lj_type].astype(np.float64)
        self.mass = s.mass.astype(np.float64)
        self.box = np.array(s.box_hi, dtype=np.float64) - np.array(s.box_lo, dtype=np.float64)
        self.rc = float(cfg.lj_cutoff)
        self.skin = float(cfg.skin)
        self.flags = torch.zeros(66, dtype=torch.int32)
        self.steps = 0
        self.list_valid = False
        self.force_evals = 0

    # ---- engine interface -----------------------------------------------------------------------
    def set_local_atoms(self, gid, ghost, pos4, vel4, lo, hi, periodic_mask):
        self.gid = gid.numpy().astype(np.int64).copy()
        self.ghost = ghost.numpy().astype(bool).copy()
        self.x = pos4.numpy()[:, :3].astype(np.float64).copy()
        self.v = vel4.numpy()[:, :3].astype(np.float64).copy()
        self.v[self.ghost] = 0.0
        self.per = [(periodic_mask >> d) & 1 for d in range(3)] if periodic_mask & 0x10 else [int(periodic_mask != 0)] * 3
        self.row = -np.ones(self.N, dtype=np.int64)
        self.row[self.gid] = np.arange(self.gid.size)
        self.f = np.zeros_like(self.x)
        self.list_valid = False
        self.n_local = self.gid.size

    def local_state(self):
        p = torch.zeros((self.n_local, 4), dtype=torch.float32)
        v = torch.zeros_like(p)
        p[:, :3] = torch.from_numpy(self.x.astype(np.float32))
        v[:, :3] = torch.from_numpy(self.v.astype(np.float32))
        return p, v

    def flag_tensor(self):
        return self.flags

    def stale_threshold(self) -> int:
        return int(np.float32((0.5 * self.skin) ** 2).view(np.uint32))

    def chunk_begin(self):
        if not self.list_valid:
            self.ref = self.x.copy()
            self.list_valid = True
        self.flags.zero_()

    def chunk_integrate(self, mode, dt, s):
        thr = self.stale_threshold()
        if int(self.flags[s]) > thr:
            if mode != 2:
                self.flags[s + 1] = max(int(self.flags[s + 1]), int(self.flags[s]))
            return
        own = ~self.ghost
        k = (dt if mode == 1 else 0.5 * dt) * ACC / self.mass[self.gid][own, None]
        self.v[own] += k * self.f[own]
        if mode != 2:
            self.x[own] += dt * self.v[own]
            d2 = float(((self.x[own] - self.ref[own]) ** 2).sum(1).max()) if own.any() else 0.0
            bits = int(np.float32(d2).view(np.uint32))
            if bits > thr:
                self.flags[s + 1] = max(int(self.flags[s + 1]), bits)

    def chunk_forces(self, s):
        if s >= 0 and int(self.flags[s + 1]) > self.stale_threshold():
            return
        if not self.list_valid:
            self.ref = self.x.copy()
            self.list_valid = True
        self.force_evals += 1
        own = np.nonzero(~self.ghost)[0]
        f = np.zeros_like(self.x)
        g = self.gid
        for i in own:
            d = self.x[i] - self.x
            for a in range(3):
                if self.per[a]:
                    d[:, a] -= np.rint(d[:, a] / self.box[a]) * self.box[a]
            r2 = (d * d).sum(1)
            m = (r2 < self.rc ** 2) & (r2 > 0)
            r2m = r2[m]
            sg = 0.5 * (self.sig[g[i]] + self.sig[g[m]])
            ep = np.sqrt(self.eps[g[i]] * self.eps[g[m]])
            s6 = (sg * sg / r2m) ** 3
            fs = 24 * ep * (2 * s6 * s6 - s6) / r2m + KE * self.q[g[i]] * self.q[g[m]] / (r2m * np.sqrt(r2m))
            f[i] = (fs[:, None] * d[m]).sum(0)
        self.f = f

    def rebuild(self):
        self.list_valid = False

    def chunk_end(self, n):
        return self.flags[:n].numpy().astype(np.uint32).copy()

    def add_steps(self, n):
        self.steps += n

    def pack(self, gid, out4, flag_word=-1):
        g = gid.numpy().astype(np.int64)
        atoms = g >= 0
        rows = self.row[g[atoms]]
        assert (rows >= 0).all()
        o = out4.numpy()
        o[atoms, :3] = self.x[rows].astype(np.float32)
        if flag_word >= 0:   # flag rows carry the bit pattern of the local flag word
            o[~atoms, 0] = np.int32(int(self.flags[flag_word])).view(np.float32)

    def unpack(self, gid, in4, shift4, flag_word=-1):
        g = gid.numpy().astype(np.int64)
        atoms = g >= 0
        rows = self.row[g[atoms]]
        assert (rows >= 0).all()
        self.x[rows] = (in4[atoms, :3] + shift4[atoms, :3]).numpy().astype(np.float64)
        if flag_word >= 0 and (~atoms).any():
            got = int(in4.numpy()[~atoms, 0].copy().view(np.int32).max())
            self.flags[flag_word] = max(int(self.flags[flag_word]), got)

    def energy(self):
        return {k: 0.0 for k in ("kinetic", "lj", "coulomb", "lj14", "coulomb14", "bond", "angle", "dihedral", "virial")}

    def stats(self):
        return {"n_atoms": self.n_local, "rebuild_count": 0}

    def profile(self, on):
        pass

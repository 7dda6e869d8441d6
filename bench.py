#!/usr/bin/env python3
"""bench.py — MD steps/s and atom-updates/s of the hot path on N MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path (velocity-Verlet kick/drift, LJ+Coulomb tile kernel, bonded
terms, rebuild trigger; neighbour rebuilds included at their natural cadence) over the synthetic
1,029,000-atom TIP3P box (BASELINE.json configs[4], "water1M"), which fits one GPU, with all
state resident in HBM before the timed region.  For N > 1 the SAME box is spatially decomposed
across the ranks (strong scaling) with ghost-atom halo exchange over RCCL.

Prints ONE JSON line on rank 0: the contract keys plus
  roofline     — dominant kernel (nb_cluster_kernel): algorithmic bytes (32 B per atom)
                 x atoms per launch / mean launch duration from HIP events on the library's stream
  cpu_baseline — oracle/cpu_production.c (fp32, half Verlet list reused across steps, OpenMP over all host cores,
                 built -O3 -march=native on the machine it runs on) on a bounded sample of the same box
The driver's command times 20 steps (10 ms: zero or one list rebuild, no energy evaluation falls into it), so after the
timed region an UNTIMED-for-`value` tail of 1000 steps runs with rebuilds at their natural cadence and energies every 100
steps; its rate is reported beside `value` as `steps_per_s_1000`.  The handle that is timed is created from the prepared
state just before the run; --settle-steps (300, untimed, reported in config.untimed_preparation) let its step loop reach
its steady state (rebuild cadence -> chunk lengths, dual-list buffer, decomposed: the split's A/B over 16 chunks) before
the W warm-up steps - with --warmup 5 alone the window measured the start-up of a new handle (tools/window_warmup.sh).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3
B_ALG_NONBONDED = 32.0         # R x(12)+q(4)+type(4), W f(12)  per atom per launch
B_ALG_INTEGRATE = 64.0         # R x,v,f (36) + 1/m (4), W x,v (24)
B_ALG_BONDED_WATER = 52.0      # 36 + 16 t, t = 1 bonded term per atom in flexible water
B_ALG_FUSED_WATER = 116.0      # bonded gather + kick + drift as one pass: the two figures above together
B_ALG_STEP_WATER = 170.0       # whole step, water box
FLOP_PER_PAIR = 45.0
def _nb_kernel_rev():
    """Revision of the default pair kernel's code (molchanica_amd/_build_info.py: comment- and white-space-insensitive hash of the
    two headers that hold the kernel + the parameter structs it reads): the cached PMC traffic figure (profiles/nb_traffic.json,
    written by tools/summarize_rocprof.py with the revision of the tree it was measured on) is quoted only while it matches."""
    from molchanica_amd._build_info import pair_kernel_rev
    return pair_kernel_rev()


NB_KERNEL_REV = _nb_kernel_rev()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="water1M", choices=["water1M", "dhfr23k", "complex50k", "dna100k"])
    ap.add_argument("--dt", type=float, default=0.0005)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=100, help="steps of the cpu_baseline leg (SURVEY 8d: 100 at the 1 M-atom box)")
    ap.add_argument("--cpu-kind", default="port-production", choices=["port-production", "port"],
                    help="port-production: oracle/cpu_production.c (fp32, half Verlet list reused across steps); port: the fp64 parity oracle")
    ap.add_argument("--reference-cmd", default="",
                    help="SURVEY 8d's slot for the real reference: a command line (no shell) that runs --reference-steps MD steps of "
                         "the same workload on the host CPU (e.g. a headless build of the reference's md loop); it is timed as a "
                         "child process and reported as cpu_baseline.kind = 'reference' instead of this repo's CPU port")
    ap.add_argument("--reference-steps", type=int, default=100, help="steps the --reference-cmd run performs")
    ap.add_argument("--tail-steps", type=int, default=-1,
                    help="untimed-for-value tail with natural rebuilds and energies every 100 steps; -1 = 1000 when --steps < 1000, else 0")
    ap.add_argument("--settle-steps", type=int, default=300,
                    help="untimed NVE steps on the handle that is timed, in front of the --warmup steps: the preparation above runs on a "
                         "handle of its own, and a new handle's step loop learns its rebuild cadence (chunk lengths), its dual-list buffer "
                         "and - decomposed - whether the interior / boundary split pays over its first ~16 chunks; with the driver's "
                         "--warmup 5 the 20 timed steps otherwise measure that start-up (0.52-0.56 ms per step against 0.48-0.52)")
    ap.add_argument("--nb-variant", type=int, default=0)
    ap.add_argument("--decomposed", action="store_true", help="drive the decomposed path even on one GPU")
    ap.add_argument("--no-equilibrate", dest="equilibrate", action="store_false",
                    help="skip the untimed relaxation + 300 K thermalisation of the synthetic start")
    ap.add_argument("--eq-min-iters", type=int, default=100)
    ap.add_argument("--eq-steps", type=int, default=600)
    ap.add_argument("--energy-every", type=int, default=100,
                    help="evaluate energies (an energy-flavoured force pass + reduction) every this many timed steps, "
                         "as SURVEY 8d's measurement contract asks; 0 = never")
    ap.add_argument("--profile-level", type=int, default=2, choices=[0, 1, 2],
                    help="HIP-event timing inside the timed region: 2 = the pair kernel only (two extra queue packets per step), "
                         "1 = every step kernel, 0 = none; bonded/integrate times always come from a short profiled tail")
    ap.add_argument("--chunk-steps", type=int, default=0, help="steps enqueued between host checks of the rebuild flag (0 = library default)")
    ap.add_argument("--skin", type=float, default=2.0, help="Verlet buffer in A (the measurement contract says 2)")
    ap.add_argument("--inner-skin", type=float, default=0.0,
                    help="dual pair list: buffer of the rolling-pruned inner list in A (0 = library default 0.5, < 0 = off)")
    ap.add_argument("--no-extras", dest="extras", action="store_false",
                    help="skip the extra keys of the default run (N = 1, water1M): `classes` = steps/s of BASELINE configs 2-4 (dhfr23k, "
                         "complex50k, dna100k) and `default_operating_point` = the reference's own configuration (rigid OPC, SPME, dt 2 fs)")
    ap.add_argument("--pme", action="store_true", help="Ewald Coulomb with the SPME reciprocal sum (not the headline config)")
    return ap.parse_args()


def cpu_baseline_production(system, cfg, dt, n_steps):
    """Times oracle/cpu_production.c (kind "port-production": this repo's fp32 restatement of a production CPU MD
    loop - the reference's Rust engine cannot be built here) on all host cores: n_steps velocity-Verlet steps of the
    same system, the half Verlet list reused until an atom has moved skin/2 (first build included)."""
    from oracle import cpu_production as cp
    lib = cp.lib(native=True)
    # cores the process may really use: the scheduler affinity, cut by a cgroup CPU quota (the GPU boxes show 256 logical CPUs
    # and grant 16 CPUs' worth of time: 128 OpenMP threads on that are 16 cores, eight-fold oversubscribed)
    usable = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(per) + 0.5))
    except Exception:
        pass
    if quota:
        usable = min(usable, quota)
    lib.cpu_prod_set_threads(int(usable))
    t0 = time.perf_counter()
    _, _, _, builds = cp.run(system, cfg, dt, n_steps, energy_every=100, native=True)
    el = time.perf_counter() - t0
    n = system.n_atoms
    cores = int(lib.cpu_prod_max_threads())
    pairs = int(lib.cpu_prod_last_pairs())
    import ctypes
    lib.cpu_prod_last_lane_pairs.restype = ctypes.c_uint64
    lanes = int(lib.cpu_prod_last_lane_pairs())
    return {
        "value": n * n_steps / el, "unit": "atom-updates/s", "steps_per_s": n_steps / el,
        "cores": cores, "kind": "port-production", "list_pairs_per_s_per_core": pairs / el / max(cores, 1),
        "lane_pairs_per_s_per_core": lanes / el / max(cores, 1),
        "cpu_quota": quota, "logical_cpus": os.cpu_count(),
        "sample": f"{n_steps} velocity-Verlet steps of the same {n}-atom box, fp32, SIMD cluster-pair loop (clusters of 8 atoms, "
                  f"structure-of-arrays, per-entry periodic shift, j-forces of an entry in registers; `omp simd`, gcc -O3 -march=native), "
                  f"half cluster-pair list (rc + skin) reused across steps ({builds} list builds incl. the first), Newton-3 with "
                  f"thread-private force buffers, energies every 100 steps, OpenMP over the {cores} CPUs this process may use "
                  f"({os.cpu_count()} logical CPUs shown, cgroup quota {quota}), {el:.1f} s: "
                  f"{pairs / el / max(cores, 1) / 1e6:.1f} M list pairs/s/core (atom pairs inside the list radius; "
                  f"{lanes / el / max(cores, 1) / 1e6:.0f} M lane pairs/s/core evaluated; list builds included in the time)",
    }


def cpu_baseline_reference(cmd, n_atoms, n_steps):
    """Times an external reference run (`--reference-cmd`): the command is started as a child process without a
    shell, and must perform n_steps MD steps of the same workload on the host cores.  Nothing in this repository produces such
    a binary (the reference's engine is a Rust crate that is not in its tree): the slot is for a maintainer who has one."""
    import shlex
    import subprocess
    argv = shlex.split(cmd)
    t0 = time.perf_counter()
    r = subprocess.run(argv, capture_output=True, text=True)
    el = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"--reference-cmd exited with {r.returncode}: {r.stderr[-400:]}")
    return {"value": n_atoms * n_steps / el, "unit": "atom-updates/s", "steps_per_s": n_steps / el,
            "cores": os.cpu_count(), "kind": "reference",
            "sample": f"{n_steps} steps by the external command {argv[0]!r} (wall time of the child process, start-up included), {el:.1f} s"}


def cpu_baseline(system, cfg, dt, n_steps):
    """Times the oracle (kind "port": this repo's C restatement — the reference's Rust engine
    cannot be built) on all host cores, on a bounded sample: n_steps velocity-Verlet steps of the
    same system (n_steps+1 force evaluations)."""
    from oracle import oracle
    try:
        import hashlib
        flags = next((l for l in open("/proc/cpuinfo") if l.startswith("flags")), "unknown")
        # one -march=native build per CPU model: such an object must never run on a machine it was not built on
        path = oracle.build(extra="-march=native", target="liborc_native_%s.so" % hashlib.sha1(flags.encode()).hexdigest()[:10])
        lib = oracle.lib(path)
    except Exception:
        lib = oracle.lib()
    import ctypes as C
    cs, cc = system.to_c(), cfg.to_c()
    n = system.n_atoms
    x = system.pos.astype(np.float64).copy()
    v = system.vel.astype(np.float64).copy()
    en = np.zeros(16)
    dp = C.POINTER(C.c_double)
    t0 = time.perf_counter()
    lib.orc_step(C.byref(cs), C.byref(cc), x.ctypes.data_as(dp), v.ctypes.data_as(dp), float(dt), int(n_steps),
                 None, en.ctypes.data_as(dp), 1)
    el = time.perf_counter() - t0
    lib.orc_max_threads.restype = C.c_int
    return {
        "value": n * n_steps / el, "unit": "atom-updates/s", "steps_per_s": n_steps / el,
        "cores": int(lib.orc_max_threads()), "kind": "port",
        "sample": f"{n_steps} velocity-Verlet steps ({n_steps + 1} force evaluations) of the same {n}-atom box, "
                  f"fp64 cell-list oracle, OpenMP over all host cores, {el:.1f} s",
    }


def with_watchdog(fn, what, rank, world, timeout_s):
    """Runs fn() (a collective of the library's own communicator: ncclCommInitRank, the transport self-test) on a helper thread.  The
    first multi-rank RCCL run happens on a box nobody watches: if the call has not returned within timeout_s - a peer that never
    arrived, a bootstrap that cannot connect - this rank says which rank, which phase and for how long, and exits non-zero
    instead of hanging until the driver's own limit.  (os._exit: the stuck call holds the GIL-free native wait, no clean way out.)"""
    import threading
    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as e:  # noqa: BLE001 - handed to the caller
            box["e"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout_s)
    if t.is_alive():
        sys.stderr.write(f"[bench rank {rank} of {world}] watchdog: {what} has not returned after {timeout_s:.0f} s - a peer did not arrive "
                         f"or the bootstrap cannot connect (MASTER_ADDR={os.environ.get('MASTER_ADDR')}, LOCAL_RANK={os.environ.get('LOCAL_RANK')}); exiting\n")
        sys.stderr.flush()
        os._exit(3)
    if "e" in box:
        raise box["e"]
    return box.get("v")


def class_rate(name, dt, device):
    """steps/s of one of BASELINE.json's smaller configs on this GPU, measured like the headline (relaxed + 300 K start on a handle of
    its own, a fresh handle, untimed settle steps, NVE with rebuilds at their natural cadence), without event brackets: ~0.3 s each."""
    import torch
    from molchanica_amd import MdConfig, systems
    from molchanica_amd.md_state import MdState
    s = systems.BY_NAME[name]()
    cfg = MdConfig()
    with MdState(s, cfg, device=device) as eq:
        eq.minimize_energy(100); eq.initialize_velocities(300.0, True, seed=105)
        eq.set_thermostat(1, 300.0, 0.02, 1); eq.step(dt, None, 600); eq.set_thermostat(0, 300.0, 0.02, 1)
        s.pos, s.vel = np.ascontiguousarray(eq.positions(), np.float32), np.ascontiguousarray(eq.velocities(), np.float32)
    n = 3000
    with MdState(s, cfg, device=device) as md:
        md.step(dt, None, 1000)                      # untimed: chunk lengths / dual-list buffer settle, the GPU is back at its clocks
        torch.cuda.synchronize()
        t0 = time.perf_counter(); md.step(dt, None, n); torch.cuda.synchronize(); el = time.perf_counter() - t0
        st = md.stats()
    return {"n_atoms": s.n_atoms, "steps": n, "steps_per_s": n / el, "ms_per_step": 1e3 * el / n,
            "atom_updates_per_s": s.n_atoms * n / el, "rebuilds_so_far": int(st["rebuild_count"]), "n_tiles": int(st["n_tiles"])}


def default_operating_point_rate(device):
    """What a user of the reference runs (/root/reference src/prefs/mod.rs:203, src/ui/panels/md.rs:362-371, README.md:236-240): rigid
    4-site OPC water (SETTLE + M virtual site), dt 2 fs, SPME, CSVR thermostat - 64^3 waters = 1,048,576 sites, skin 2 A."""
    import torch
    from molchanica_amd import MdConfig, systems
    from molchanica_amd.md_state import MdState
    s = systems.opc_water_box(64, seed=5)
    cfg = MdConfig(coulomb_mode=2, ewald_alpha=0.3, overrides=0, skin=2.0)
    n = 500
    with MdState(s, cfg, device=device) as md:
        md.minimize_energy(100); md.initialize_velocities(300.0, True, seed=1)
        md.set_thermostat(1, 300.0, 0.02, 1); md.step(0.001, None, 1500)       # untimed: the random-orientation lattice relaxes
        md.set_thermostat(2, 300.0, 0.1, 10, seed=2); md.step(0.002, None, 300)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); md.step(0.002, None, n); e = md.energy(); torch.cuda.synchronize(); el = time.perf_counter() - t0
        st = md.stats()
    return {"workload": "opc64: 262,144 rigid OPC waters = 1,048,576 sites, SPME (order 4, ~1 A mesh, beta 0.3), rc 10 A, skin 2 A, "
                        "dt 2 fs, CSVR 300 K every 10 steps; 500 timed steps + one energy read",
            "n_sites": s.n_atoms, "steps": n, "steps_per_s": n / el, "ms_per_step": 1e3 * el / n, "ns_per_day": n / el * 0.002e-3 * 86400,
            "temperature_K": round(float(e["temperature"]), 1), "rebuilds_so_far": int(st["rebuild_count"])}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    # Verification aid for 1-GPU boxes: MDX_BENCH_SAME_GPU=1 maps every rank to device 0, uses gloo for the launcher's process
    # group and the library's shared-memory transport instead of RCCL (which refuses two ranks on one device), so the
    # complete process-per-rank flow - rendezvous, broadcast of the prepared state, mdx_comm_init*, the decomposed step loop
    # with its halo exchange, repartition, energy reduction, the JSON line - runs on real kernels.  Never a measurement.
    same_gpu = os.environ.get("MDX_BENCH_SAME_GPU", "0") == "1"
    if same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # The launcher's group only carries the prepared state, the 128-byte RCCL id and the timing reduction (the library's
        # own communicator is initialised separately).  The backend is chosen BEFORE the one init call and alike on every
        # rank: from MDX_BENCH_PG_BACKEND if set, else "nccl" when torch was built with it, else "gloo".  (A second
        # init_process_group after a failed one cannot work - the env:// rendezvous is consumed - and a per-rank fallback
        # could leave the ranks on different backends.)
        backend = os.environ.get("MDX_BENCH_PG_BACKEND", "")
        if not backend:
            backend = "gloo" if (same_gpu or not dist.is_nccl_available()) else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    pg_cpu = same_gpu or (world > 1 and dist.get_backend() == "gloo")    # tensors of the launcher's collectives live on the host

    from molchanica_amd import MdConfig, systems
    from molchanica_amd.md_state import MdState

    system = systems.BY_NAME[args.workload]()
    cfg = MdConfig(nb_variant=args.nb_variant, skin=args.skin, chunk_steps=args.chunk_steps, inner_skin=args.inner_skin)  # rc 10 Å (LJ & Coulomb), skin 2 Å, shifted cutoff Coulomb
    if args.pme:
        cfg = MdConfig(nb_variant=args.nb_variant, coulomb_mode=2, ewald_alpha=0.3, overrides=0, inner_skin=args.inner_skin)
    n_atoms = system.n_atoms

    # Untimed preparation of the synthetic box (SURVEY 8d asks for Maxwell-Boltzmann at 300 K): the
    # generator places waters on a jittered lattice with random orientations, and run as-is that
    # potential energy heats the box to ~1300 K.  So: steepest-descent relaxation, velocities drawn at
    # 300 K, a short Berendsen-coupled run, thermostat off.  The timed region is plain NVE.  Rank 0
    # prepares, everyone receives the same state.
    prep = None
    if args.equilibrate and system.periodic:
        t_prep = time.perf_counter()
        if rank == 0:
            with MdState(system, cfg, device=local_rank) as eq:
                eq.minimize_energy(args.eq_min_iters)
                eq.initialize_velocities(300.0, True, seed=105)
                eq.set_thermostat(1, 300.0, 0.02, 1)
                eq.step(args.dt, None, args.eq_steps)
                eq.set_thermostat(0, 300.0, 0.02, 1)
                e_eq = eq.energy()
                pos_eq, vel_eq = eq.positions(), eq.velocities()
        else:
            pos_eq = np.zeros((n_atoms, 3), np.float32); vel_eq = np.zeros((n_atoms, 3), np.float32)
            e_eq = {"temperature": 0.0}
        if world > 1:
            tp = torch.from_numpy(np.ascontiguousarray(pos_eq)); tv = torch.from_numpy(np.ascontiguousarray(vel_eq))
            if not pg_cpu:
                tp, tv = tp.cuda(), tv.cuda()
            dist.broadcast(tp, 0); dist.broadcast(tv, 0)
            pos_eq, vel_eq = tp.cpu().numpy(), tv.cpu().numpy()
        system.pos = np.ascontiguousarray(pos_eq, dtype=np.float32)
        system.vel = np.ascontiguousarray(vel_eq, dtype=np.float32)
        prep = {"min_iters": args.eq_min_iters, "thermostat_steps": args.eq_steps,
                "temperature_K": round(float(e_eq["temperature"]), 1), "seconds": round(time.perf_counter() - t_prep, 2)}

    md = MdState(system, cfg, device=local_rank)
    stepper = lambda k: md.step(args.dt, None, k)
    stats = md.stats
    prof = md.profile
    parallelism = "single"
    if world > 1 or args.decomposed:
        # The decomposed step loop lives below the C ABI (include/mdx.h, mdx_comm_init): rank 0 draws the RCCL id, the
        # launcher's process group only carries those 128 bytes; halo exchange, stale-list protocol, repartition and
        # the energy all-reduce are the library's own RCCL calls.
        from molchanica_amd.md_state import comm_unique_id
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(comm_unique_id()), dtype=torch.uint8).clone()
        if world > 1:
            uid = uid if pg_cpu else uid.cuda()
            dist.broadcast(uid, 0)
            uid = uid.cpu()
        transport = "RCCL"
        shm_name = "bench_" + bytes(uid.numpy().tobytes())[:8].hex()
        # MDX_BENCH_TRY_RCCL=1 (with MDX_BENCH_SAME_GPU=1): go through the RCCL branch all the same - on one device ncclCommInitRank
        # refuses the second rank, which exercises exactly the failure -> shared-memory fallback path below.
        # MDX_BENCH_FAIL_RANK=k (tests): rank k never calls mdx_comm_init - the peers' watchdogs must fire.
        comm_timeout = float(os.environ.get("MDX_BENCH_COMM_TIMEOUT_S", "120"))
        if same_gpu and world > 1 and os.environ.get("MDX_BENCH_TRY_RCCL", "0") != "1":
            md.comm_init_shm(shm_name, rank, world)
            transport = "shared memory (verification aid, not a measurement)"
        else:
            # RCCL below the C ABI has only ever run with one rank per box before the driver's scaling run.  If its
            # initialisation or its self-test (one halo-shaped exchange + one all-reduce) fails on ANY rank, every rank
            # says so on stderr and the run continues over the library's host shared-memory transport - GPU kernels
            # unchanged, halo through host memory - and the JSON line names the transport it was measured on.
            err = ""
            if os.environ.get("MDX_BENCH_FAIL_RANK", "") == str(rank):
                sys.stderr.write(f"[bench rank {rank}] MDX_BENCH_FAIL_RANK: this rank stays away from mdx_comm_init (fault injection)\n"); sys.stderr.flush()
                time.sleep(10.0 * comm_timeout)
                raise SystemExit(4)
            try:
                with_watchdog(lambda: md.comm_init(bytes(uid.numpy().tobytes()), rank, world), "mdx_comm_init (ncclCommInitRank + first partition)", rank, world, comm_timeout)
                with_watchdog(md.comm_selftest, "mdx_comm_selftest (send/recv group to every rank + all-reduces)", rank, world, comm_timeout)
            except Exception as e:  # noqa: BLE001 - reported, never swallowed
                err = f"{type(e).__name__}: {e}"
            if world > 1:
                bad = torch.tensor([1.0 if err else 0.0], device="cpu" if pg_cpu else "cuda")
                with_watchdog(lambda: dist.all_reduce(bad, op=dist.ReduceOp.MAX), "the launcher's all-reduce of the transport verdict", rank, world, comm_timeout)
                if bad.item() > 0:
                    sys.stderr.write(f"[bench rank {rank} of {world}, device {local_rank}] RCCL transport unusable: "
                                     f"{err or 'no error on this rank (another rank failed)'}; continuing over the shared-memory transport\n")
                    sys.stderr.flush()
                    md.close()
                    md = MdState(system, cfg, device=local_rank)
                    md.comm_init_shm(shm_name, rank, world)
                    transport = "host shared memory (RCCL initialisation failed - see stderr)"
            elif err:
                raise SystemExit(err)
        stepper = lambda k: md.step(args.dt, None, k)
        stats = md.stats
        prof = md.profile
        info = md.comm_info()
        g = info["grid"]
        parallelism = (f"spatial {g[0]}x{g[1]}x{g[2]} bricks, ghost halo {info['halo']:.1f} A, ncclSend/ncclRecv group per step "
                       f"(below the C ABI), interior tiles beside the message when that measures faster, stale flag on the halo message, "
                       f"local list rebuilds, then repartition; transport: {transport}")

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run_steps(n):
        """n steps with energies every --energy-every steps -> number of energy evaluations."""
        n_e = 0
        if args.energy_every > 0:
            # energies at the steps whose absolute count is a multiple of the cadence - the steps the library was told about
            # (mdx_set_energy_cadence: they are evaluated with the forces of those steps, not by a second evaluation)
            done = 0
            while done < n:
                k = min(args.energy_every - md.step_count % args.energy_every, n - done)
                stepper(k); done += k
                if md.step_count % args.energy_every == 0:
                    md.energy(); n_e += 1
        else:
            stepper(n)
        return n_e

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device="cpu" if pg_cpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    if args.energy_every > 0:
        md.set_energy_cadence(args.energy_every)
    if args.settle_steps > 0:
        stepper(args.settle_steps)
    stepper(args.warmup)
    if world > 1 and args.profile_level == 2:
        # decomposed runs are timed in their production arrangement: the event brackets would switch the interior /
        # boundary split of the pair kernel off (its halves run on two streams); kernel times come from the profiled tail
        args.profile_level = 0
    prof(args.profile_level)
    st0 = stats()
    sync()
    t0 = time.perf_counter()
    n_energy = run_steps(args.steps)
    sync()
    el = max_over_ranks(time.perf_counter() - t0)
    st = stats()
    # Untimed-for-`value` tail: the driver's 20-step window holds no list rebuild and no energy evaluation, so the
    # representative rate (natural rebuild cadence, energies every 100 steps) is measured here and reported beside it.
    tail_steps = args.tail_steps if args.tail_steps >= 0 else (1000 if args.steps < 1000 else 0)
    tail = None
    if tail_steps > 0:
        sync()
        t1 = time.perf_counter()
        n_energy_tail = run_steps(tail_steps)
        sync()
        el_tail = max_over_ranks(time.perf_counter() - t1)
        st_t = stats()
        tail = {"steps": tail_steps, "steps_per_s": tail_steps / el_tail, "ms_per_step": 1e3 * el_tail / tail_steps,
                "rebuilds": int(st_t["rebuild_count"] - st["rebuild_count"]), "energy_evaluations": n_energy_tail,
                "rebuild_ms_per_step_amortised": (st_t["rebuild_ms_sum"] - st["rebuild_ms_sum"]) / tail_steps,
                "prune_passes": int(st_t.get("prune_passes", 0) - st.get("prune_passes", 0))}
        st_nb = st_t                       # pair-kernel launch statistics: timed region + tail
    else:
        st_nb = st
    # bonded / integrate kernel times: a short tail outside the timed region with every kernel bracketed
    prof(1)
    st_pre = stats()
    stepper(48)
    sync()
    st_tail = stats()
    prof(0)
    # the large classes run bonded gather + kick + drift as ONE pass on 15 steps of a 16-step chunk: its own timer
    fused_n = st_tail.get("fused_launches", 0) - st_pre.get("fused_launches", 0)
    fused_ms = (st_tail.get("fused_ms_sum", 0.0) - st_pre.get("fused_ms_sum", 0.0)) / fused_n if fused_n else None
    streaming_ms_per_step = ((st_tail["bonded_ms_sum"] - st_pre["bonded_ms_sum"]) + (st_tail["integ_ms_sum"] - st_pre["integ_ms_sum"])
                             + (st_tail.get("fused_ms_sum", 0.0) - st_pre.get("fused_ms_sum", 0.0))) / 48.0
    if args.profile_level == 0:
        st_nb = st_tail

    # Decomposed runs: where the step time goes, rank by rank (mdx_comm_diag): 48 more steps with every phase of the step
    # bracketed in its production arrangement (profile level 3), outside the timed region.
    multi = None
    if world > 1 or args.decomposed:
        prof(3)
        t_d = time.perf_counter()
        stepper(48)
        sync()
        wall_d = (time.perf_counter() - t_d) / 48 * 1e3
        mine = md.comm_diag()
        prof(0)
        mine["step_wall_ms_profiled"] = wall_d
        mine["rebuild_fallbacks"] = int(stats().get("rebuild_fallbacks", 0))
        mine["phase_ms_per_step"] = {k: v / 48.0 for k, v in mine.pop("phase_ms_sum").items()}
        gpu_sum = sum(mine["phase_ms_per_step"].values())
        mine["gpu_phases_ms_per_step"] = gpu_sum
        mine["host_and_idle_ms_per_step"] = wall_d - gpu_sum      # (phases on two streams overlap when the split is on: may be negative)
        per_rank = [mine]
        if world > 1:
            per_rank = [None] * world
            dist.all_gather_object(per_rank, mine)
        multi = {"rccl_world": mine["rccl_comm_count"], "rccl_version": mine["rccl_version"], "transport": mine["transport"],
                 "halo_bytes_per_step_per_rank": [r["halo_bytes_per_step"] for r in per_rank],
                 "n_owned": [r["n_owned"] for r in per_rank], "n_ghost": [r["n_ghost"] for r in per_rank],
                 "repartitions": [r["repartitions"] for r in per_rank], "local_rebuilds": [r["local_rebuilds"] for r in per_rank],
                 "rebuild_fallbacks": [r["rebuild_fallbacks"] for r in per_rank],
                 "overlap_split_kept": [r["overlap_split"] for r in per_rank],
                 # one message per step (full shell) or two (half shell + force return): chosen below the ABI from the message time
                 # measured when the handles joined (include/mdx.h: mdx_comm_diag.wire_ns_measured)
                 "half_shell": [r["half_shell"] for r in per_rank], "wire_us_measured_at_attach": [r["wire_ns_measured"] / 1e3 for r in per_rank],
                 "phase_ms_per_step": {k: [round(r["phase_ms_per_step"][k], 5) for r in per_rank] for k in mine["phase_ms_per_step"]},
                 "step_wall_ms_profiled": [round(r["step_wall_ms_profiled"], 5) for r in per_rank],
                 "host_and_idle_ms_per_step": [round(r["host_and_idle_ms_per_step"], 5) for r in per_rank],
                 "note": "GPU time between HIP events per phase, 48 untimed steps at profile level 3 (production arrangement); "
                         "halo_wire / force_wire = the ncclSend/ncclRecv group incl. waiting for the peers; pair = whole launch or interior half"}

    steps_per_s = args.steps / el
    value = n_atoms * steps_per_s
    # Whether one of the ~0.6 ms list rebuilds (one per ~25 steps) lands in a 20-step timed window is a coin flip worth 5 % of
    # `value`.  Two figures beside it that do not depend on the coin: the rate with the window's rebuild time taken out, and
    # that rate with the rebuild cost of a long run (the tail's, per step) put back in.
    rb_ms_timed = st["rebuild_ms_sum"] - st0["rebuild_ms_sum"]
    ms_no_rb = (1e3 * el - rb_ms_timed) / args.steps
    nb_launches = st_nb["nb_launches"]
    nb_ms = st_nb["nb_ms_sum"] / max(nb_launches, 1)
    bonded_ms = st_tail["bonded_ms_sum"] / max(st_tail["bonded_launches"], 1)
    integ_ms = st_tail["integ_ms_sum"] / max(st_tail["integ_launches"], 1)
    atoms_per_launch = st["n_atoms"]                 # owned + ghost atoms of this rank's launch
    gbs = lambda b_per_atom, ms: b_per_atom * atoms_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    achieved = gbs(B_ALG_NONBONDED, nb_ms)
    # (i-cluster, j-cluster) pairs x 8 x 8 lanes.  With the dual list most step-loop launches walk the inner list (mean
    # over the pruning passes) and a fraction `prune_frac` walks the Verlet list while it re-prunes.
    n_steps_nb = max(int(st_nb["step_count"] - st0["step_count"]), 1)
    prune_frac = min(1.0, (st_nb.get("prune_passes", 0) - st0.get("prune_passes", 0)) / n_steps_nb)
    dual = st_nb.get("prune_passes", 0) > 0 and st_nb["n_inner_cluster_pairs"] > 0
    verlet_evals = float(st_nb["n_cluster_pairs"]) * 64
    inner_evals = float(st_nb["n_inner_cluster_pairs"]) * 64 if dual else verlet_evals
    pair_evals = (1.0 - prune_frac) * inner_evals + prune_frac * verlet_evals if dual else verlet_evals
    # algorithmic flops: 45 per half pair INSIDE the cutoff; their number follows from the density (homogeneous box,
    # exclusions neglected: 209.4 per atom for water at rc = 10 A) - only priced for the plain cutoff flavour
    alg_tflops = None
    if system.periodic and not args.pme and world == 1:
        vol = float(np.prod(np.asarray(system.box_hi, np.float64) - np.asarray(system.box_lo, np.float64)))
        half_pairs_per_atom = 0.5 * (4.0 / 3.0) * np.pi * max(cfg.lj_cutoff, cfg.coulomb_cutoff) ** 3 * n_atoms / vol
        alg_tflops = FLOP_PER_PAIR * half_pairs_per_atom * n_atoms / (nb_ms * 1e-3) / 1e12 if nb_ms > 0 else 0.0
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "nb_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            # a CACHED figure: PMC passes cannot run inside the bench.  Only quoted for the kernel/flavour it was
            # collected on; a stale or foreign entry is nulled rather than repeated.
            if (tj.get("workload") == args.workload and world == 1 and not args.pme and args.nb_variant in (0, 5)
                    and tj.get("kernel_rev") == NB_KERNEL_REV):
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = f"profiles/nb_traffic.json ({tj.get('source', 'rocprofv3 --pmc')}; cached, not measured by this run)"
        except Exception:
            pass
    if args.nb_variant == 1:
        kernel = "nb_tile_kernel"
    elif dual:
        kernel = (f"nb_cluster_kernel: {100 * (1 - prune_frac):.0f} % inner-list walks + {100 * prune_frac:.0f} % pruning passes "
                  f"(one merged launch per step: the device picks the body)")
    else:
        kernel = "nb_cluster_kernel (plain Verlet list)"
    out = {
        "metric": "MD atom-updates/sec (and steps/sec), 1M-atom solvated box",
        "value": value, "unit": "atom-updates/s", "steps_per_s": steps_per_s,
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * el / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,   # the ONE 1M-atom box at every N (BASELINE.json)
        "dtype": "f32", "data": "synthetic",
        "value_no_rebuild": n_atoms * 1e3 / ms_no_rb if (ms_no_rb > 0 and args.profile_level) else None,
        "value_with_amortised_rebuild": (n_atoms * 1e3 / (ms_no_rb + (tail["rebuild_ms_per_step_amortised"] if tail else rb_ms_timed / args.steps))
                                         if (ms_no_rb > 0 and args.profile_level) else None),
        "steps_per_s_1000": tail["steps_per_s"] if tail else steps_per_s,
        "rebuild_ms_per_step_amortised": (tail["rebuild_ms_per_step_amortised"] if tail
                                          else (st["rebuild_ms_sum"] - st0["rebuild_ms_sum"]) / args.steps),
        "tail": tail,
        "config": {"workload": args.workload, "n_atoms": n_atoms, "lj_cutoff": cfg.lj_cutoff,
                   "coulomb_cutoff": cfg.coulomb_cutoff, "skin": cfg.skin, "dt_ps": args.dt,
                   "coulomb": "ewald real space + SPME (order 4, ~1 A mesh)" if args.pme else "shifted cutoff", "parallelism": parallelism,
                   "rebuilds_in_timed_region": int(st["rebuild_count"] - st0["rebuild_count"]),
                   "dual_list": ({"inner_skin": cfg.inner_skin or 0.5, "verlet_pair_evals": verlet_evals, "inner_pair_evals": inner_evals,
                                  "prune_frac": prune_frac} if dual else None),
                   "energy_evaluations_in_timed_region": n_energy,
                   "untimed_preparation": dict(prep or {}, nve_settle_steps_on_the_timed_handle=args.settle_steps),
                   "repartitions": int(st_nb["repartitions"]) if (world > 1 or args.decomposed) else None,
                   "local_rebuilds": int(st_nb["local_rebuilds"]) if (world > 1 or args.decomposed) else None,
                   "rebuild_fallbacks": int(st_nb.get("rebuild_fallbacks", 0)),
                   "n_owned_rank0": int(st_nb["n_owned"]), "n_ghost_rank0": int(st_nb["n_ghost"])},
        "roofline": {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "launch_ms": nb_ms, "launches": nb_launches,
                     "launch_sampling": ("HIP events around every %s-th executed pair launch of the timed region + tail (MDX_PROF_SAMPLE; a bracket on "
                                         "every launch costs 10-14 us of a step: 2 %% at water1M, 28 %% at dhfr23k)" % os.environ.get("MDX_PROF_SAMPLE", "8")),
                     "algorithmic_bytes_per_launch": B_ALG_NONBONDED * atoms_per_launch,
                     "note": "pair loop is fp32-VALU bound, see valu.frac; HBM fraction is low by physics"},
        # the streaming kernels, in SURVEY 8d's algorithmic bytes (profiled 48-step tail, every kernel bracketed)
        "roofline_streaming": {
            "integrate_kernel": {"bound": "hbm", "bytes_per_atom": B_ALG_INTEGRATE, "launch_ms": integ_ms,
                                 "achieved": gbs(B_ALG_INTEGRATE, integ_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": gbs(B_ALG_INTEGRATE, integ_ms) / HBM_PEAK_GBS},
            "bonded_gather_kernel": ({"bound": "hbm", "bytes_per_atom": B_ALG_BONDED_WATER, "launch_ms": bonded_ms,
                                      "achieved": gbs(B_ALG_BONDED_WATER, bonded_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": gbs(B_ALG_BONDED_WATER, bonded_ms) / HBM_PEAK_GBS}
                                     if args.workload == "water1M" else {"launch_ms": bonded_ms, "achieved": None}),
            # steps 1 .. 15 of a 16-step chunk: bonded gather + full kick + drift in one pass (mdx_integrate.hip); the two
            # kernels above then run once per chunk (the opening half kick, the chunk's last bonded call, the closing kick)
            "bonded_integrate_kernel": ({"bound": "hbm", "bytes_per_atom": B_ALG_FUSED_WATER, "launch_ms": fused_ms,
                                         "achieved": gbs(B_ALG_FUSED_WATER, fused_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": gbs(B_ALG_FUSED_WATER, fused_ms) / HBM_PEAK_GBS, "launches_of_48_steps": fused_n}
                                        if (fused_ms and args.workload == "water1M") else ({"launch_ms": fused_ms} if fused_ms else None))},
        "valu": {"pair_evals_per_launch": pair_evals,
                 "pair_evals_per_s": pair_evals / (nb_ms * 1e-3) if nb_ms > 0 else 0.0,
                 "algorithmic_tflops": alg_tflops, "peak_tflops": FP32_PEAK_TFLOPS,
                 "frac": alg_tflops / FP32_PEAK_TFLOPS if alg_tflops is not None else None},
        "step_hbm_frac": B_ALG_STEP_WATER * value / (world * HBM_PEAK_GBS * 1e9) if args.workload == "water1M" else None,
        "multi_gpu": multi,
        "kernel_ms": {"nonbonded": nb_ms, "bonded": bonded_ms, "integrate": integ_ms, "bonded_integrate_fused": fused_ms,
                      "bonded_plus_integrate_per_step": streaming_ms_per_step,
                      "rebuild_total": st_nb["rebuild_ms_sum"] - st0["rebuild_ms_sum"]},
    }
    if rank == 0 and world == 1 and args.extras and args.workload == "water1M" and not args.pme and not args.decomposed:
        # BASELINE configs 2-4 and the reference's own operating point, measured by whoever runs this command (the driver), on this
        # GPU, after the headline numbers are in hand; a failure is reported inside the line, never instead of it
        md.close()
        out["classes"] = {}
        for w in ("dhfr23k", "complex50k", "dna100k"):
            try:
                out["classes"][w] = class_rate(w, args.dt, local_rank)
            except Exception as e:  # noqa: BLE001
                out["classes"][w] = {"steps_per_s": None, "error": f"{type(e).__name__}: {e}"}
        try:
            out["default_operating_point"] = default_operating_point_rate(local_rank)
        except Exception as e:  # noqa: BLE001
            out["default_operating_point"] = {"steps_per_s": None, "error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the GPU measurement above must reach the driver whatever happens to the CPU leg (a failed -march=native build,
        # a missing compiler on the box): its failure is reported inside the JSON line, not instead of it
        try:
            if args.reference_cmd:
                out["cpu_baseline"] = cpu_baseline_reference(args.reference_cmd, n_atoms, args.reference_steps)
            elif args.cpu_kind == "port-production" and not args.pme:
                out["cpu_baseline"] = cpu_baseline_production(system, cfg, args.dt, args.cpu_steps)
            else:
                out["cpu_baseline"] = cpu_baseline(system, cfg, args.dt, min(args.cpu_steps, 10))
        except Exception as e:  # noqa: BLE001
            out["cpu_baseline"] = {"value": None, "unit": "atom-updates/s", "cores": os.cpu_count(), "kind": args.cpu_kind,
                                   "sample": None, "error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""What ONE rank of an N-GPU run has to do per step, measured alone on the GPU: rank 0 of the N-rank decomposition of the
1M-atom box, driven by the library's own decomposed step loop (mdx_comm_init_null: the production path with a transport
that delivers nothing - ghosts keep their positions, reductions are the identity).  Kernel and host costs per step are
those of a real rank; only the wire time of the halo message is missing.
Usage: timeout 300 python tools/one_rank_profile.py [world] [steps=100]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from molchanica_amd import systems, MdConfig
from molchanica_amd.md_state import MdState

worlds = [int(sys.argv[1])] if len(sys.argv) > 1 else [1, 2, 4, 8]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
s = systems.water1m() if "ONE_RANK_NSIDE" not in os.environ else systems.water_box(int(os.environ["ONE_RANK_NSIDE"]))   # a small box shows the host-side floor per step
if os.environ.get("ONE_RANK_HOT", "0") != "1":      # same untimed preparation as bench.py: relaxed, 300 K
    with MdState(s, MdConfig()) as eq:
        eq.minimize_energy(100); eq.initialize_velocities(300.0, True, seed=105)
        eq.set_thermostat(1, 300.0, 0.02, 1); eq.step(0.0005, None, 600); eq.set_thermostat(0, 300.0, 0.02, 1)
        s.pos = np.ascontiguousarray(eq.positions(), dtype=np.float32); s.vel = np.ascontiguousarray(eq.velocities(), dtype=np.float32)
# Half-shell halo: a lone rank gets no ghost forces back (its lower neighbours, which evaluate those pairs, do not exist), so the
# atoms at its lower faces feel half a neighbourhood and the box is unphysical after ~70 steps.  Every measurement stretch
# therefore starts from a FRESH handle on the prepared state and is kept to <= 48 steps (natural rebuild cadence: ~2 per stretch).
STRETCH = min(steps, 48)


def fresh(world, overlap):
    # arrangements of the decomposed step: "pipe" (round 5: one pair launch whose workgroups draw tiles, both messages on the compute
    # stream), "1" = "split" (rounds 2-4: interior tiles on a side stream, two launches), "0" = plain chain, "auto" = the library tries
    os.environ.pop("MDX_HALO_PIPE", None)
    if overlap == "auto": os.environ.pop("MDX_HALO_OVERLAP", None)      # the library tries both and keeps the faster
    elif overlap == "pipe": os.environ["MDX_HALO_OVERLAP"] = "1"; os.environ["MDX_HALO_PIPE"] = "1"
    elif overlap in ("1", "split"): os.environ["MDX_HALO_OVERLAP"] = "1"
    else: os.environ["MDX_HALO_OVERLAP"] = "0"
    md = MdState(s, MdConfig())
    md.comm_init_null(0, world)
    md.step(0.0005, None, 4)
    return md


# ONE_RANK_WIRE="0,25": MDX_NULL_WIRE_US per arm - every send/recv group of the null transport then occupies its stream for that many
# microseconds and the unpack / add kernels run (on loop-back buffers): the step of a real rank with a stated wire time.  Unset: the
# transport of rounds 3-4 (nothing enqueued for a message, unpack and add skipped).
wires = os.environ.get("ONE_RANK_WIRE", "").split(",")
for world in worlds:
  for wire in wires:
    if wire == "": os.environ.pop("MDX_NULL_WIRE_US", None)
    else: os.environ["MDX_NULL_WIRE_US"] = wire
    for overlap in (os.environ.get("ONE_RANK_SPLIT", "pipe,1,0").split(",")):      # ONE_RANK_SPLIT=pipe / 1 / 0: one arm only
          n_rep = max(1, steps // STRETCH)
          wall = 0.0; rebuilds = 0; reparts = 0; rb_sum = 0.0; rp_sum = 0.0; rb_n = 0; rp_n = 0; fallbacks = 0
          for rep in range(n_rep):                      # plain (event-free) stretches: the step wall
              with fresh(world, overlap) as md:
                  md.profile(2); md.profile(0)          # (resets the timers)
                  st0 = md.stats()
                  t0 = time.perf_counter(); md.step(0.0005, None, STRETCH); st1 = md.stats(); wall += time.perf_counter() - t0    # stats() synchronises
                  rebuilds += st1["rebuild_count"] - st0["rebuild_count"]; reparts += st1["repartitions"] - st0["repartitions"]
                  fallbacks += st1.get("rebuild_fallbacks", 0) - st0.get("rebuild_fallbacks", 0)
          n_steps = n_rep * STRETCH
          if os.environ.get("ONE_RANK_TRACE", "0") == "1":     # under rocprofv3: stop here, the trace ends with plain (event-free) steps
              print("world %d wire %s arrangement %s: step wall %.3f ms" % (world, wire or "-", overlap, 1e3 * wall / n_steps)); continue
          for rep in range(n_rep):                      # the same stretches with the rebuild timer on: what the list builds cost
              with fresh(world, overlap) as md:
                  md.profile(2); r0 = md.stats()
                  md.step(0.0005, None, STRETCH); r1 = md.stats(); md.profile(0)
                  rb_sum += r1["rebuild_ms_sum"] - r0["rebuild_ms_sum"]; rb_n += r1["rebuild_count"] - r0["rebuild_count"]
                  rp_sum += r1["repartition_ms_sum"] - r0["repartition_ms_sum"]; rp_n += r1["repartitions"] - r0["repartitions"]
          rb_ms = rb_sum / max(rb_n, 1); rp_ms = rp_sum / max(rp_n, 1)
          with fresh(world, overlap) as md:             # every kernel bracketed
              md.profile(1)
              md.step(0.0005, None, 48)
              st = md.stats()
              md.profile(0)
          k_nb = st["nb_ms_sum"] / max(st["nb_launches"], 1); k_b = st["bonded_ms_sum"] / max(st["bonded_launches"], 1)
          k_i = st["integ_ms_sum"] / max(st["integ_launches"], 1)
          amort = (rebuilds * rb_ms + reparts * max(rp_ms - rb_ms, 0.0)) / n_steps
          print("world %d rank 0 (wire %s us per message, arrangement %s): owned %d ghost %d tiles %d | pair %.3f ms bonded %.3f integrate %.3f = %.3f ms of kernels | "
                "step wall %.3f ms (%d list rebuilds in %d steps at %.2f ms - %d left the fused chain -, %d repartitions at %.2f ms incl. their rebuild: %.3f ms per step amortised; %.3f ms of the wall is in neither kernels nor list builds) -> ceiling %.0f steps/s without wire time | cluster pairs verlet %.1f M inner %.1f M" % (
                    world, wire or "-", {"1": "split", "0": "plain", "pipe": "pipelined"}.get(overlap, "auto-tuned"), st["n_owned"], st["n_ghost"], st["n_tiles"], k_nb, k_b, k_i, k_nb + k_b + k_i,
                    1e3 * wall / n_steps, rebuilds, n_steps, rb_ms, fallbacks, reparts, rp_ms, amort,
                    1e3 * wall / n_steps - (k_nb + k_b + k_i) - amort, n_steps / wall,
                    st["n_cluster_pairs"] / 1e6, st["n_inner_cluster_pairs"] / 1e6), flush=True)

"""Copies the judged artefacts of tools/final_evidence.sh from gpurun_out/TAG_final (+ gpurun_out/TAG: tools/profile_round.sh) into profiles/
under the round's name.  Usage: python tools/publish_final.py r06"""
import os, shutil, sys
tag = sys.argv[1]
F, P = os.path.join("gpurun_out", tag + "_final"), os.path.join("gpurun_out", tag)
def put(src, dst, head=None):
    if not os.path.exists(src):
        print("missing:", src); return
    dst = os.path.join("profiles", dst)
    if head is None: shutil.copy(src, dst)
    else: open(dst, "w").write(head + open(src).read())
    print(dst)
put(os.path.join(P, "bench.json"), f"{tag}_bench.json")
put(os.path.join(P, f"{tag}_rocprof_summary.txt"), f"{tag}_rocprof_summary.txt")
put(os.path.join(P, f"{tag}_nb_pmc_raw.json"), f"{tag}_nb_pmc_raw.json")
put(os.path.join(P, "nb_traffic.json"), "nb_traffic.json")
put(os.path.join(F, "bench_driver_cmd.json"), f"{tag}_bench_driver_cmd.json")
for w in ("dhfr23k", "complex50k", "dna100k"):
    put(os.path.join(F, f"bench_{w}.json"), f"{tag}_bench_{w}.json")
put(os.path.join(F, "one_rank.txt"), f"{tag}_one_rank_of_N.txt",
    "# tools/one_rank_profile.py N 192: rank 0 of N of water1M alone on one MI355X (null transport; 'wire x us per message' = MDX_NULL_WIRE_US, '-' = nothing\n"
    "# enqueued for a message).  MDX_HALF_SHELL=1: two messages per step (ghost forces travel back), =0: one message, cross pairs on both ranks;\n"
    "# last block: MDX_HALF_SHELL unset - the handle measures the message time when it joins and chooses (mdx_dd_attach).\n")
dp = ""
for f, what in (("dp64.txt", "tools/default_point_time.py 64 (skin 2 A)"), ("dp18.txt", "tools/default_point_time.py 18 (skin 2 A)"),
                ("dp64_skin_auto.txt", "DP_ONLY=spme tools/default_point_time.py 64 0 0 (mdx_config.skin = 0: the library chooses)")):
    p = os.path.join(F, f)
    if os.path.exists(p): dp += f"# {what}\n" + open(p).read()
open(os.path.join("profiles", f"{tag}_default_operating_point.txt"), "w").write(
    "# The reference's default operating point (rigid OPC, SPME, CSVR, dt 2 fs; /root/reference src/prefs/mod.rs:203, src/ui/panels/md.rs:362-371) on one MI355X\n" + dp)
kt = ""
for f, what in (("dp_serial.txt", "MDX_PME_OVERLAP=0: the reciprocal-space chain on the handle's stream (stand-alone kernel durations, rooflines in the bytes each kernel moves)"),
                ("dp_overlap.txt", "default arrangement: the chain on its side stream beside the pair kernel")):
    p = os.path.join(F, f)
    if os.path.exists(p): kt += f"# --- {what}\n" + open(p).read()
open(os.path.join("profiles", f"{tag}_rocprof_summary_default_point.txt"), "w").write(
    f"# Kernel trace of the reference's default operating point (1,048,576 OPC sites, SPME, dt 2 fs), round {tag[1:]} HEAD: tools/kt_default_point_single.sh\n" + kt)
pm = open(os.path.join(F, "parity_margins.txt")).read() if os.path.exists(os.path.join(F, "parity_margins.txt")) else ""
c5 = open(os.path.join(F, "c5_outliers.txt")).read() if os.path.exists(os.path.join(F, "c5_outliers.txt")) else ""
open(os.path.join("profiles", f"{tag}_parity_margins.txt"), "w").write(
    "# tests/parity_margins.py: what the parity tests' tolerances leave in hand, with the worst atom of every config explained\n" + pm +
    "# tools/dbg/c5_outliers.py: water1M after 32 steps of the step loop (the 1300 K state), step-loop and plain-list forces against the oracle\n" + c5)
put(os.path.join(F, "single_point.txt"), f"{tag}_single_point_latency.txt")
put(os.path.join(F, "nve_soak.txt"), f"{tag}_nve_conservation.txt")
put(os.path.join(F, "decomp_soak.txt"), f"{tag}_decomp_soak.txt")
if os.path.exists(os.path.join(F, "gputest.log")):
    tail = [l for l in open(os.path.join(F, "gputest.log")).read().splitlines() if "passed" in l or "failed" in l]
    open(os.path.join("profiles", f"{tag}_gputest_summary.txt"), "w").write("\n".join(tail) + "   # python -m pytest tests -m gpu -q at HEAD, one MI355X (gpurun)\n")

"""Copies the judged artefacts of a profile round from gpurun_out/TAG into profiles/ under the round's
name.  Usage: python tools/publish_profiles.py r01c r01"""
import os, shutil, sys
tag, rnd = sys.argv[1], sys.argv[2]
src = os.path.join("gpurun_out", tag)
for a, b in [(f"{tag}_rocprof_summary.txt", f"{rnd}_rocprof_summary.txt"), (f"{tag}_nb_pmc_raw.json", f"{rnd}_nb_pmc_raw.json"),
             ("bench.json", f"{rnd}_bench.json"), ("nb_traffic.json", "nb_traffic.json")]:
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join("profiles", b))
        print("profiles/" + b)

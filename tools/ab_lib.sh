#!/bin/bash
# A/B of builds of the library on the headline workload: tools/ab_lib.sh <tag> libA.so libB.so ...   (names under molchanica_amd/)
# Each build runs twice, interleaved; prints steps/s and the HIP-event kernel times of bench.py.
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for rep in 1 2; do
  for L in "$@"; do
    MDX_LIB=$PWD/molchanica_amd/$L python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline $AB_ARGS > gpurun_out/$TAG/ab_${L}_$rep.json 2> gpurun_out/$TAG/ab_${L}_$rep.err
    python3 - "$L" gpurun_out/$TAG/ab_${L}_$rep.json <<'PY'
import json, sys
j = json.load(open(sys.argv[2]))
print(sys.argv[1], round(j["steps_per_s"], 1), {k: (round(v, 4) if v is not None else None) for k, v in j["kernel_ms"].items()})
PY
  done
done

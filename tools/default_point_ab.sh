#!/bin/bash
# the reference's default operating point (1 M sites, SPME): stand-alone kernel times with their rooflines (MDX_PME_OVERLAP=0),
# then steps/s with the reciprocal chain on its side stream, plain and with a high-priority stream: bash tools/default_point_ab.sh TAG
TAG=${1:-dpab}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"
MDX_PME_OVERLAP=0 bash tools/kt_default_point_single.sh ${TAG}_serial 64 > "$OUT/serial.txt" 2>&1
python3 tools/default_point_time.py 64 2>&1 | grep SPME > "$OUT/overlap.txt"
MDX_PME_PRIORITY=1 python3 tools/default_point_time.py 64 2>&1 | grep SPME > "$OUT/overlap_priority.txt"
echo "== MDX_PME_OVERLAP=0 (stand-alone kernel times)"; cat "$OUT/serial.txt"; echo "== side stream"; cat "$OUT/overlap.txt"; echo "== side stream, high priority"; cat "$OUT/overlap_priority.txt"

#!/bin/bash
# exact pruning at water1M: one wave per tile (default from 12 k tiles) against eight: bash tools/prune_ab.sh TAG
TAG=${1:-prab}; export TMPDIR=/tmp
for mw in 12000 30000; do
  OUT=$PWD/gpurun_out/${TAG}_$mw; mkdir -p "$OUT"
  MDX_PRUNE_MW_BELOW=$mw rocprofv3 --kernel-trace -d "$OUT/kt" -o kt -- python3 bench.py --steps 200 --warmup 30 --no-cpu-baseline --tail-steps 0 > "$OUT/bench.json" 2> "$OUT/kt.err"
  echo "MDX_PRUNE_MW_BELOW=$mw"; python3 tools/rebuild_timeline.py "$OUT/kt" | grep -i "prune\|build_list\|kernels,"
  find "$OUT" -name "*.db" -delete
done

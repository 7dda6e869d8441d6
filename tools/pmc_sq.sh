#!/bin/bash
# SQ counters of the step kernels only (two passes).  Usage via gpurun: bash tools/pmc_sq.sh TAG
set -u
TAG=${1:-sq}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d "$OUT/pmc_SQ" -o pmc -- python3 bench.py --steps 24 --warmup 4 --no-cpu-baseline > "$OUT/bench_pmc_SQ.json" 2> "$OUT/pmc_SQ.err"
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_BRANCH -d "$OUT/pmc_SQ2" -o pmc -- python3 bench.py --steps 24 --warmup 4 --no-cpu-baseline > "$OUT/bench_pmc_SQ2.json" 2> "$OUT/pmc_SQ2.err"
python3 tools/summarize_rocprof.py "$OUT" "$OUT" "$TAG"
cat "$OUT/${TAG}_rocprof_summary.txt"
find "$OUT" -name "*.db" -size +20M -delete

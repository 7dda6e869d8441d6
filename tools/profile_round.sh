#!/bin/bash
# Collects the round's evidence on the GPU box: bench line, rocprofv3 kernel-trace stats and
# (in separate passes, as the MI355X guide prescribes) the HBM traffic counters of the pair kernel.
# Usage (from the repo root, through gpurun):  bash tools/profile_round.sh r01
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 1000 --warmup 100 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 3000 "$OUT/bench.json"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras > "$OUT/bench_kt.json" 2> "$OUT/kt.err"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o pmc -- python3 bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras > "$OUT/bench_pmc_$C.json" 2> "$OUT/pmc_$C.err"
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d "$OUT/pmc_SQ" -o pmc -- python3 bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras > "$OUT/bench_pmc_SQ.json" 2> "$OUT/pmc_SQ.err"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$OUT/pmc_SQ2" -o pmc -- python3 bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras > "$OUT/bench_pmc_SQ2.json" 2> "$OUT/pmc_SQ2.err"
python3 tools/summarize_rocprof.py "$OUT" "$OUT" "$TAG" > /dev/null
find "$OUT" -name "*.db" -delete        # the summaries stay, the raw databases do not travel back
du -sh "$OUT"

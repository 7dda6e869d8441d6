#!/bin/bash
# One call that collects the round's evidence on the GPU box: bash tools/final_evidence.sh TAG   (outputs under gpurun_out/TAG_final/)
# Round 6: the driver's own command first (its JSON line carries the classes and the default operating point), then the 1000-step line
# with the rocprofv3 summary and the PMC traffic of the pair kernel (profiles/nb_traffic.json is keyed on the kernel's code: run this LAST).
TAG=${1:-r06}; F=$PWD/gpurun_out/${TAG}_final; mkdir -p "$F"; export TMPDIR=/tmp
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$F/bench_driver_cmd.json" 2> "$F/bench_driver_cmd.err"
bash tools/profile_round.sh $TAG > "$F/profile_round.log" 2>&1
for w in dhfr23k complex50k dna100k; do python3 bench.py --workload $w --steps 3000 --warmup 200 --no-cpu-baseline > "$F/bench_$w.json" 2> /dev/null; done
for hs in 1 0; do for n in 2 4 8; do MDX_HALF_SHELL=$hs ONE_RANK_WIRE=",0,25" ONE_RANK_SPLIT="1,0" python3 tools/one_rank_profile.py $n 192 2>/dev/null | grep "^world" | sed "s/^/MDX_HALF_SHELL=$hs /" >> "$F/one_rank.txt"; done; done
for n in 8; do ONE_RANK_WIRE="0,5,10,25,50" ONE_RANK_SPLIT="0" python3 tools/one_rank_profile.py $n 192 2>/dev/null | grep "^world" | sed "s/^/shell chosen by the measured wire time: /" >> "$F/one_rank.txt"; done
python3 tools/default_point_time.py 64 > "$F/dp64.txt" 2>/dev/null
python3 tools/default_point_time.py 18 > "$F/dp18.txt" 2>/dev/null
DP_ONLY=spme python3 tools/default_point_time.py 64 0 0 > "$F/dp64_skin_auto.txt" 2>/dev/null
MDX_PME_OVERLAP=0 bash tools/kt_default_point_single.sh ${TAG}_final_serial 64 > "$F/dp_serial.txt" 2>&1
bash tools/kt_default_point_single.sh ${TAG}_final_overlap 64 > "$F/dp_overlap.txt" 2>&1
python3 tests/parity_margins.py > "$F/parity_margins.txt" 2>&1
python3 tools/dbg/c5_outliers.py > "$F/c5_outliers.txt" 2>/dev/null
python3 tools/single_point_time.py > "$F/single_point.txt" 2>&1
python3 tools/nve_soak.py > "$F/nve_soak.txt" 2>&1
python3 tools/decomp_soak.py > "$F/decomp_soak.txt" 2>&1
python3 -m pytest tests -m gpu -q > "$F/gputest.log" 2>&1
tail -3 "$F/gputest.log"

# A/B of the longest-lists-first tile order (MDX_TILE_LPT: 0 off, 1 on, unset = the library's default) for mid-size launches:
# one rank of 8 / 4 / 2 of the 1 M-atom box, and the single-device workloads.  Usage via gpurun: bash tools/lpt_ab.sh
for l in 0 default; do
  if [ $l = default ]; then unset MDX_TILE_LPT; else export MDX_TILE_LPT=$l; fi
  echo "== MDX_TILE_LPT=$l"
  for w in 8 4 2; do ONE_RANK_SPLIT=0 python3 tools/one_rank_profile.py $w 192 2>&1 | grep "^world" | cut -c1-330; done
  for w in dhfr23k complex50k dna100k water1M; do python3 bench.py --workload $w --steps 2000 --warmup 200 --no-cpu-baseline --tail-steps 0 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$w', round(d['steps_per_s'],1), d['kernel_ms']['nonbonded'])
"; done
  ONE_RANK_NSIDE=40 ONE_RANK_SPLIT=0 python3 tools/one_rank_profile.py 1 192 2>&1 | grep "^world" | cut -c1-330
done

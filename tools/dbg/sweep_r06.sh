for rep in 1 2; do
for a in "--inner-skin 0.4" "--inner-skin 0.5" "--inner-skin 0.6" "--inner-skin 0.7" "--chunk-steps 24" "--chunk-steps 32" "--chunk-steps 8"; do
  python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-extras --tail-steps 0 $a 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a', round(d['steps_per_s'],1), round(d['kernel_ms']['nonbonded'],4), round(d['rebuild_ms_per_step_amortised'],4), d['config']['rebuilds_in_timed_region'], d['config']['dual_list']['prune_frac'])
"
done; done

for s in 0.25 0.3 0.35 0.4 0.5 0.65; do
  python3 bench.py --steps 1000 --warmup 100 --no-cpu-baseline --tail-steps 0 --inner-skin $s 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$s', round(d['steps_per_s'],1), d['kernel_ms']['nonbonded'], d['config']['dual_list'])
"
done

for m in 1 0; do for w in water1M dhfr23k; do
MDX_DUAL_MERGED=$m python bench.py --workload $w --no-cpu-baseline --tail-steps 0 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('merged=$m', '$w', round(j['steps_per_s'],1), 'steps/s  nb_ms', round(j['kernel_ms']['nonbonded'],4))"
done; done

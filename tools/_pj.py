import sys, json
for l in sys.stdin:
    if l.startswith("{"):
        j = json.loads(l); print(sys.argv[1], j["steps_per_s"], j["kernel_ms"]["nonbonded"])

"""The first steps of a NEW handle, launch by launch (bench.py --settle-steps 0 under rocprofv3 --kernel-trace): for the last
`n` executed pair launches of the run, the wall time from one to the next and every other kernel in between - where a fresh
handle's start-up goes.  Usage (through gpurun): rocprofv3 --kernel-trace -d gpurun_out/fh -o fh -- python3 bench.py --gpus 1
--steps 60 --warmup 5 --settle-steps 0 --no-cpu-baseline --tail-steps 0 ; python3 tools/fresh_handle_timeline.py gpurun_out/fh 70"""
import glob, os, re, sqlite3, sys, collections
db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 70
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda s: (re.match(r"(?:void )?([A-Za-z0-9_]+)", s) or re.search(r"(.*)", s)).group(1)
nb = [i for i, r in enumerate(rows) if "nb_cluster_kernel" in r[0] and r[2] - r[1] > 50e3]
sel = nb[-n:]
for a, b in zip(sel[:-1], sel[1:]):
    other = collections.OrderedDict()
    for r in rows[a + 1:b]:
        k = short(r[0]); d = other.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (r[2] - r[1]) / 1e3
    busy = sum(v[1] for v in other.values()) + (rows[a][2] - rows[a][1]) / 1e3
    wall = (rows[b][1] - rows[a][1]) / 1e3
    print(f"pair {(rows[a][2] - rows[a][1]) / 1e3:7.1f} us  wall to next {wall:8.1f}  idle {wall - busy:7.1f} | " +
          " ".join(f"{k}x{v[0]}:{v[1]:.0f}" for k, v in other.items() if k not in ("bonded_integrate_kernel",) or v[0] != 1))

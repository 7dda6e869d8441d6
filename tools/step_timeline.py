"""Timeline of one rank's step loop from a rocprofv3 kernel trace: for the last N steps of the run, every kernel with its
start offset, duration and the idle gap in front of it - where the step's wall time goes besides the kernels.
Usage (through gpurun):
  rocprofv3 --kernel-trace -d gpurun_out/tl -o tl -- python3 tools/one_rank_profile.py 8 60
  python3 tools/step_timeline.py gpurun_out/tl"""
import glob, os, re, sqlite3, sys
src = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: (re.match(r"(?:void )?([A-Za-z0-9_]+)", n) or re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:<|\(|$)", n) or re.search(r"(.*)", n)).group(1)
# steps are delimited by integrate_kernel launches; take a stretch without list rebuilds near the end
idx = [i for i, r in enumerate(rows) if short(r[0]) in ("integrate_kernel", "bonded_integrate_kernel")]
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
stretch = None
# STEP_TL_END_FRAC: search backwards from this fraction of the run (bench.py ends with profiled, unfused tails)
for end in range(int((len(idx) - 1) * float(os.environ.get('STEP_TL_END_FRAC', '1.0'))), n_steps, -1):
    seg = rows[idx[end - n_steps]:idx[end]]
    if not any(short(r[0]) in ("build_list_kernel", "bin_atoms_kernel", "kinetic_kernel", "rb_prep_kernel") for r in seg):
        stretch = (idx[end - n_steps], idx[end]); break
a, b = stretch
t0 = rows[a][1]
tot_k = tot_gap = 0.0
per = {}
prev_end = rows[a - 1][2]
for name, st, en in rows[a:b]:
    k = short(name)
    gap, dur = (st - prev_end) / 1e3, (en - st) / 1e3
    d = per.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += dur; d[2] += max(gap, 0.0)
    tot_k += dur; tot_gap += max(gap, 0.0); prev_end = max(prev_end, en)
wall = (rows[b][1] - rows[a][1]) / 1e3
print(f"{n_steps} steps without a list rebuild: wall {wall / n_steps:.1f} us per step = kernels {tot_k / n_steps:.1f} + idle gaps {tot_gap / n_steps:.1f} (overlapping kernels count once in wall)")
for k, d in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:40s} {d[0] / n_steps:5.2f} launches/step  {d[1] / n_steps:7.2f} us/step  idle gap in front {d[2] / n_steps:6.2f} us/step")

if len(sys.argv) > 3:      # every kernel of the first two steps of the stretch
    for name, st, en in rows[a:idx[idx.index(a) + 2]]:
        print(f"    +{(st - t0) / 1e3:8.1f} us  dur {(en - st) / 1e3:7.1f}  {short(name)}")

if len(sys.argv) > 4:      # what surrounds the steps: the next N kernels behind the stretch with the idle gap in front of each (chunk ends, stale lists)
    prev = rows[b - 1][2]
    for name, st, en in rows[b:b + int(sys.argv[4])]:
        print(f"    after  +{(st - rows[b][1]) / 1e3:8.1f} us  gap {(st - prev) / 1e3:6.1f}  dur {(en - st) / 1e3:7.1f}  {short(name)}")
        prev = max(prev, en)

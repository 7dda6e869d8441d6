#!/bin/bash
# kernel trace of tools/class_steps.py (no event brackets) and the timeline of one list rebuild in it: bash tools/kt_class_steps.sh TAG [WORKLOAD]
TAG=${1:-ktcs}; WL=${2:-dhfr23k}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt_$WL" -o kt -- python3 tools/class_steps.py $WL 600 > "$OUT/run_$WL.log" 2> "$OUT/kt_$WL.err"
TIMELINE_BEFORE=6 python3 tools/rebuild_timeline.py "$OUT/kt_$WL" -3 > "$OUT/timeline_$WL.txt" 2>&1
cat "$OUT/timeline_$WL.txt"
python3 - "$OUT/kt_$WL" <<'PY'
import glob, os, sqlite3, sys, re
db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.match(r"(?:void )?([A-Za-z0-9_]+)", n).group(1)
# steady steps: consecutive executed pair launches (> 10 us) with nothing but each other in between
d = [(b - a) / 1e3 for n, a, b in rows if short(n) == "nb_cluster_kernel" and b - a > 10000]
gaps = [(rows[i + 1][1] - rows[i][2]) / 1e3 for i in range(len(rows) - 1) if short(rows[i][0]) == "nb_cluster_kernel" and short(rows[i + 1][0]) == "nb_cluster_kernel" and rows[i][2] - rows[i][1] > 10000]
d.sort(); gaps.sort()
print(f"executed pair launches: n {len(d)} median {d[len(d)//2]:.1f} us, p10 {d[len(d)//10]:.1f}, p90 {d[9*len(d)//10]:.1f}; gap between consecutive ones: median {gaps[len(gaps)//2]:.2f} us (n {len(gaps)})")
PY
find "$OUT" -name "*.db" -delete

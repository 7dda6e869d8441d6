#!/bin/bash
# kernel trace of one rank of 8 at the reference's default operating point (tools/default_point_one_rank.py): the last stretch's kernels
TAG=${1:-kt_dp}; W=${2:-8}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 tools/default_point_one_rank.py $W > "$OUT/run.log" 2> "$OUT/kt.err"
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys, re
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end, grid_x from kernels order by start").fetchall()
# the slab stretches come third from the end (4 stretches slab, then 4 replicated): take the window of the 3rd quarter of the last 2/5
t0, t1 = rows[0][1], rows[-1][2]
lo, hi = t0 + 0.62 * (t1 - t0), t0 + 0.80 * (t1 - t0)
st = {}
for n, a, b, g in rows:
    if a < lo or a > hi: continue
    k = re.match(r"(?:void )?([A-Za-z0-9_]+)", n).group(1)
    d = st.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (b - a) / 1e3
tot = sum(v[1] for v in st.values())
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k:44s} n={v[0]:5d} total_us={v[1]:10.1f} avg_us={v[1]/v[0]:8.2f} {100*v[1]/tot:5.1f}%")
PY
tail -3 "$OUT/run.log"
find "$OUT" -name "*.db" -size +20M -delete

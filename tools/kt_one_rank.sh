#!/bin/bash
# kernel trace of one rank of N (tools/one_rank_profile.py, plain steps only): bash tools/kt_one_rank.sh TAG WORLD
TAG=${1:-kt1}; W=${2:-8}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
export ONE_RANK_TRACE=1 ONE_RANK_SPLIT=${ONE_RANK_SPLIT:-0}
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 tools/one_rank_profile.py $W 96 > "$OUT/run.log" 2> "$OUT/kt.err"
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys, re
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the last 60 % of the trace: the measured stretches (the preparation of the box comes first)
t_lo = rows[0][1] + 0.6 * (rows[-1][2] - rows[0][1])
st = {}
for n, a, b in rows:
    if a < t_lo: continue
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)(<[^>]*>)?", n)
    k = m.group(1) + ((m.group(2) or "").replace(" ", "") if m.group(1).startswith("nb_") else "")
    d = st.setdefault(k, [0, 0.0, 0.0]); d[0] += 1; d[1] += (b - a) / 1e3; d[2] = max(d[2], (b - a) / 1e3)
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{k:70s} n={v[0]:5d} total_us={v[1]:10.1f} avg_us={v[1]/v[0]:8.2f} max_us={v[2]:8.2f}")
PY
cat "$OUT/run.log" | tail -2
find "$OUT" -name "*.db" -size +20M -delete

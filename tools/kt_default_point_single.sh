#!/bin/bash
# kernel trace of the reference's default operating point on one GPU (tools/default_point_time.py N_SIDE): bash tools/kt_default_point_single.sh TAG [N_SIDE=64]
TAG=${1:-kt_dps}; NS=${2:-64}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 tools/default_point_time.py $NS > "$OUT/run.log" 2> "$OUT/kt.err"
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys, re
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# the SPME run comes first (minimise, 1500 + 300 untimed steps, 500 timed): take the window 30 % .. 48 % of the trace = timed SPME steps
t0, t1 = rows[0][1], rows[-1][2]
lo, hi = t0 + 0.30 * (t1 - t0), t0 + 0.48 * (t1 - t0)
st = {}
for n, a, b in rows:
    if a < lo or a > hi: continue
    k = re.match(r"(?:void )?([A-Za-z0-9_]+)", n).group(1)
    d = st.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (b - a) / 1e3
tot = sum(v[1] for v in st.values())
nstep = max(1, st.get("nb_cluster_kernel", [1])[0])
print(f"window {1e-6*(hi-lo):.1f} ms, {nstep} pair launches, kernel time per pair launch {tot/nstep:.1f} us, wall per pair launch {1e-3*(hi-lo)/nstep:.1f} us")
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:26]:
    print(f"{k[:60]:60s} n={v[0]:5d} total_us={v[1]:10.1f} avg_us={v[1]/v[0]:8.2f} per_step={v[1]/nstep:7.2f} {100*v[1]/tot:5.1f}%")
PY
tail -3 "$OUT/run.log"
find "$OUT" -name "*.db" -size +20M -delete

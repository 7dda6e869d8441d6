"""GPU idle time between consecutive kernels of a rocprofv3 kernel trace, bucketed by gap length.
Usage (on the GPU box, after tools/kt.sh-style tracing with the .db kept): python tools/gap_analysis.py DB"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# steady part only: from the 60 % mark on (past preparation and warm-up)
rows = rows[int(len(rows) * 0.6):]
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
buckets = {}
prev_end, prev_name = rows[0][2], rows[0][0]
for name, s, e in rows[1:]:
    g = s - prev_end
    if g > 0:
        key = "<2us" if g < 2e3 else "<5us" if g < 5e3 else "<10us" if g < 1e4 else "<30us" if g < 3e4 else "<100us" if g < 1e5 else "<300us" if g < 3e5 else ">=300us"
        b = buckets.setdefault(key, [0, 0.0, {}])
        b[0] += 1; b[1] += g
        k = re.match(r"(?:void )?([A-Za-z0-9_]+)", prev_name).group(1) + " -> " + re.match(r"(?:void )?([A-Za-z0-9_]+)", name).group(1)
        b[2][k] = b[2].get(k, 0.0) + g
    prev_end, prev_name = max(prev_end, e), name
print(f"span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms ({100*(span-busy)/span:.1f} %)  kernels {len(rows)}")
for key in ["<2us", "<5us", "<10us", "<30us", "<100us", "<300us", ">=300us"]:
    if key in buckets:
        n, t, ks = buckets[key]
        top = sorted(ks.items(), key=lambda kv: -kv[1])[:3]
        print(f"  gaps {key:8s} n={n:6d} total {t/1e6:8.3f} ms   top: " + "; ".join(f"{k} {v/1e6:.2f}" for k, v in top))

#!/bin/bash
# kernel trace of tools/repartition_cost.py and the timeline of one repartition: bash tools/kt_repartition.sh TAG [WORLD=8]
TAG=${1:-ktrp}; W=${2:-8}; OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d "$OUT/kt" -o kt -- python3 tools/repartition_cost.py $W 48 > "$OUT/run.log" 2> "$OUT/kt.err"
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys, re
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.match(r"(?:void )?([A-Za-z0-9_]+)", n).group(1)
idx = [i for i, r in enumerate(rows) if short(r[0]) == "dd_classify_kernel"]
print(len(idx), "classify launches")
if idx:
    i0 = idx[-1]
    while i0 > 0 and short(rows[i0 - 1][0]) not in ("nb_cluster_kernel", "bonded_gather_kernel", "integrate_kernel", "bonded_integrate_kernel"): i0 -= 1
    t0 = rows[i0][1]; prev = rows[i0 - 1][2]
    for name, a, b in rows[i0:i0 + 48]:
        k = short(name)
        print(f"  +{(a - t0) / 1e3:8.1f} us  gap {(a - prev) / 1e3:6.1f}  dur {(b - a) / 1e3:7.1f}  {k}")
        prev = max(prev, b)
        if k.startswith("nb_cluster") and (b - a) > 30e3: break
PY
cat "$OUT/run.log"
find "$OUT" -name "*.db" -delete

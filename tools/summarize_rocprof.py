"""Summarises rocprofv3's rocpd SQLite outputs (kernel trace + PMC passes) into small text/JSON
files for profiles/.  Usage: python tools/summarize_rocprof.py gpurun_out/r01 profiles r01"""
import glob
import json
import os
import re
import sqlite3
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name


lines = []
# ---- kernel trace stats -------------------------------------------------------------------------
kt = glob.glob(os.path.join(src, "kt", "*.db"))
stats = {}
if kt:
    db = sqlite3.connect(kt[0])
    rows = db.execute("select name, start, end from kernels").fetchall()
    for name, st, en in rows:
        d = stats.setdefault(short(name), [0, 0.0, 1e30, 0.0])
        dur = (en - st) / 1e3
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    total = sum(v[1] for v in stats.values())
    lines.append(f"# rocprofv3 --kernel-trace --stats  ({tag}; python3 bench.py --steps 300 --warmup 30 --no-cpu-baseline)")
    lines.append(f"{'kernel':44s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k:44s} {v[0]:7d} {v[1]:12.1f} {v[1] / v[0]:10.2f} {v[2]:10.2f} {v[3]:10.2f} {100 * v[1] / total:6.2f}")
    lines.append("")
    lines.append("note: pair-kernel launches that were gated off behind a stale neighbour list (no-ops, a few us)")
    lines.append("      are included in 'calls'; see the per-kernel avg of launches > 100 us below.")
    big = [(en - st) / 1e3 for name, st, en in rows if short(name).startswith("nb_") and (en - st) > 100e3]
    if big:
        lines.append(f"nb pair kernel, executed launches only: n={len(big)} avg_us={sum(big) / len(big):.2f} "
                     f"min_us={min(big):.2f} max_us={max(big):.2f}")

# ---- PMC passes ----------------------------------------------------------------------------------
pmc = {}
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    dbs = glob.glob(os.path.join(d, "*.db"))
    if not dbs or not os.path.isdir(d):
        continue
    db = sqlite3.connect(dbs[0])
    try:
        cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
        rows = db.execute("select * from counters_collection").fetchall()
    except Exception as e:  # pragma: no cover
        lines.append(f"{d}: {e}")
        continue
    ci = {c: i for i, c in enumerate(cols)}
    kname = "kernel_name" if "kernel_name" in ci else ("name" if "name" in ci else None)
    cname = "counter_name" if "counter_name" in ci else "pmc_name"
    vname = "value" if "value" in ci else "counter_value"
    if kname is None:
        lines.append(f"{d}: unknown schema {cols}")
        continue
    dur_ok = "start" in ci and "end" in ci
    for r in rows:
        k = short(r[ci[kname]])
        if dur_ok and k.startswith("nb_") and (r[ci["end"]] - r[ci["start"]]) < 100e3:
            continue   # gated no-op launch
        e = pmc.setdefault(k, {}).setdefault(r[ci[cname]], [0, 0.0])
        e[0] += 1; e[1] += float(r[ci[vname]])
if pmc:
    lines.append("")
    lines.append(f"# rocprofv3 --pmc (separate passes; mean per executed dispatch)   ({tag})")
    for k, cs in sorted(pmc.items()):
        if not (k.startswith("nb_") or k.startswith("bonded") or k.startswith("integrate")):
            continue
        lines.append(k)
        for c, (n, tot) in sorted(cs.items()):
            lines.append(f"    {c:28s} n={n:5d} mean={tot / n:18.1f}")

open(os.path.join(dst, f"{tag}_rocprof_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# ---- HBM traffic of the pair kernel (MI355X_MICROARCH.md §HBM: FETCH_SIZE on gfx950 reports 1/2 of the
# bytes of wide coalesced reads -> x2; units are KiB-ish 'FETCH_SIZE'=KB, WRITE_SIZE uncalibrated) ----
nbk = [k for k in pmc if k.startswith("nb_")]
if nbk:
    cs = pmc[nbk[0]]
    out = {"kernel": nbk[0], "raw": {c: v[1] / v[0] for c, v in cs.items()}}
    json.dump(out, open(os.path.join(dst, f"{tag}_nb_pmc_raw.json"), "w"), indent=1)

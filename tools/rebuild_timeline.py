"""Timeline of ONE list rebuild out of a rocprofv3 kernel trace: every kernel between the first kernel of a rebuild (rb_prep /
unsort / rebuild_clear) and the pair kernel that follows it, with its start offset, duration and the idle gap in front of it.
Usage: python tools/rebuild_timeline.py gpurun_out/TAG/kt [which=-2]   (which: index of the rebuild in the trace, default the last but one)"""
import glob, os, re, sqlite3, sys
src = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
db = sqlite3.connect(glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
short = lambda n: re.match(r"(?:void )?([A-Za-z0-9_]+)", n).group(1)
starts = [i for i, r in enumerate(rows) if short(r[0]) in ("rb_prep_kernel", "rebuild_clear_kernel")]
if not starts: sys.exit("no rebuild in the trace")
i0 = starts[which]
while i0 > 0 and short(rows[i0 - 1][0]) in ("unsort_kernel",): i0 -= 1
t0 = rows[i0][1]; prev_end = rows[i0 - 1][2] if i0 else t0
print(f"rebuild #{which} of {len(starts)}: gap in front of its first kernel {(t0 - prev_end) / 1e3:.1f} us (previous kernel: {short(rows[i0 - 1][0]) if i0 else '-'})")
nb = int(os.environ.get("TIMELINE_BEFORE", "0"))      # kernels in front of the rebuild (the step loop noticing the stale list)
for j in range(max(0, i0 - nb), i0):
    name, a, b = rows[j]
    print(f"  {(a - t0) / 1e3:9.1f} us  gap {(a - rows[j - 1][2]) / 1e3 if j else 0.0:6.1f}  dur {(b - a) / 1e3:7.1f}  {short(name)}")
tot_k = 0.0; n = 0
for name, a, b in rows[i0:i0 + 60]:
    k = short(name)
    print(f"  +{(a - t0) / 1e3:8.1f} us  gap {(a - prev_end) / 1e3:6.1f}  dur {(b - a) / 1e3:7.1f}  {k}")
    prev_end = b; tot_k += (b - a) / 1e3; n += 1
    if k.startswith("nb_") and n > 3: break
print(f"  {n} kernels, {tot_k:.1f} us of kernel time in {(prev_end - t0) / 1e3:.1f} us")

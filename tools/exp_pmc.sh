#!/bin/bash
# SQ counters of the pair kernel on the frozen water1M state (tools/exp_split_cost.py): bash tools/exp_pmc.sh TAG [lib.so]
# (MDX_LIB is exported here, before rocprofv3 starts python3 itself)
set -u
TAG=${1:-pmc}; LIB=${2:-}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
[ -n "$LIB" ] && export MDX_LIB=$PWD/molchanica_amd/$LIB
[ -f /tmp/exp_state_water1M.npz ] || MDX_LIB= python3 tools/exp_split_cost.py prep
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_BRANCH"
P3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P -d "$OUT/pmc_$i" -o pmc -- python3 tools/exp_split_cost.py time $TAG > "$OUT/run_$i.json" 2> "$OUT/run_$i.err"
done
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys, collections
out = sys.argv[1]
res = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    for f in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
        db = sqlite3.connect(f)
        try:
            rows = db.execute("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection group by dispatch_id, counter_name").fetchall()
        except Exception as e:
            tabs = [r[0] for r in db.execute("select name from sqlite_master").fetchall()]
            print("no counters_collection view in", f, e, tabs[:20]); continue
        acc = {}
        for k, c, disp, v in rows:
            if "nb_cluster_kernel" not in k: continue
            acc.setdefault(c, []).append(v)
        for c, vs in acc.items():
            vs = sorted(vs)
            big = [v for v in vs if v > 0.2 * vs[-1]]     # executed launches (the gated-off twin counts almost nothing)
            res[c] = (sum(big) / max(len(big), 1), len(big), len(vs))
for c, (v, n, m) in res.items():
    print(f"{c:28s} {v:16.1f}   (mean of {n} executed launches of {m})")
PY
cat "$OUT"/run_1.json
find "$OUT" -name "*.db" -size +20M -delete

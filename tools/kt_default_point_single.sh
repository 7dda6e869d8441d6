#!/bin/bash
# kernel trace of the reference's default operating point on one GPU (tools/default_point_time.py N_SIDE): bash tools/kt_default_point_single.sh TAG [N_SIDE=64]
TAG=${1:-kt_dps}; NS=${2:-64}
OUT=$PWD/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
export DP_ONLY=${DP_ONLY:-spme}
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o kt -- python3 tools/default_point_time.py $NS > "$OUT/run.log" 2> "$OUT/kt.err"
python3 - "$OUT" <<'PY'
import glob, os, sqlite3, sys, re
out = sys.argv[1]
db = sqlite3.connect(glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
# DP_ONLY=spme makes tools/default_point_time.py run the SPME case alone: its last 500 steps are the timed ones, i.e. everything
# from the 500th-last executed pair launch on (the thermostatted relaxation at dt 1 fs in front has a different rebuild cadence)
nb = [i for i, r in enumerate(rows) if "nb_cluster_kernel" in r[0] and (r[2] - r[1]) > 50e3]
i_lo = nb[-500] if len(nb) > 500 else 0
lo, hi = rows[i_lo][1], rows[-1][2]
st = {}
for n, a, b in rows:
    if a < lo or a > hi: continue
    k = re.match(r"(?:void )?([A-Za-z0-9_]+)", n).group(1)
    d = st.setdefault(k, [0, 0.0]); d[0] += 1; d[1] += (b - a) / 1e3
tot = sum(v[1] for v in st.values())
nstep = 500
print(f"window {1e-6*(hi-lo):.1f} ms, {nstep} pair launches, kernel time per pair launch {tot/nstep:.1f} us, wall per pair launch {1e-3*(hi-lo)/nstep:.1f} us")
for k, v in sorted(st.items(), key=lambda kv: -kv[1][1])[:26]:
    print(f"{k[:60]:60s} n={v[0]:5d} total_us={v[1]:10.1f} avg_us={v[1]/v[0]:8.2f} per_step={v[1]/nstep:7.2f} {100*v[1]/tot:5.1f}%")
# per-kernel HBM roofline of the reciprocal-space chain in ALGORITHMIC bytes (only meaningful for stand-alone times: MDX_PME_OVERLAP=0)
ns = int(os.environ.get("DP_NSIDE", "64")); sites = 4 * ns ** 3; charged = 3 * ns ** 3
K = int(os.environ.get("DP_MESH", "200")); pts = K ** 3; cplx = K * K * (K // 2 + 1)
def row(name, key, nbytes, note):
    if key not in st: return
    us = st[key][1] / st[key][0]
    print(f"roofline {name:34s} {us:8.1f} us  {nbytes / 1e6:8.1f} MB algorithmic  {nbytes / us / 1e6:7.2f} TB/s = {100 * nbytes / us / 1e6 / 8.0:5.1f} % of 8 TB/s   ({note})")
fk = [k for k in st if k.startswith("fft_rtc")]
print(f"# {sites} sites, {charged} charged, mesh {K}^3 = {pts} points")
row("pme_spread_tile_kernel", "pme_spread_tile_kernel", charged * (16 + 64 * 4), "16 B posq + 4^3 x 4 B mesh RMW per charge")
row("pme_bin_kernel", "pme_bin_kernel", sites * 17 + charged * 20, "posq + flag per slot, 20-B record per charge")
row("pme_canvas_kernel", "pme_canvas_kernel", charged * 20 + pts * 4 * 1.67, "20-B record per charge in, canvases out: (19/16)^3 x mesh")
row("pme_combine_kernel", "pme_combine_kernel", pts * 4 * (1.67 + 1.0), "canvases in, mesh out")
row("pme_gather_kernel", "pme_gather_kernel", charged * (16 + 64 * 4) + sites * 16, "16 B posq + 4^3 x 4 B mesh reads per charge + 16 B force row per slot")
row("pme_gather_brick_kernel", "pme_gather_brick_kernel", charged * (28 + 16) + pts * 4 * 1.67, "28-B record in + 16-B force row out per charge, the potential once per covering canvas: (19/16)^3 x mesh")
row("pme_solve_kernel", "pme_solve_kernel", cplx * (4 + 16), "theta 4 B + complex RW 16 B per point of the half-complex mesh")
pitch = (K // 2 + 1 + 15) // 16 * 16
row("pme_xpass_solve_kernel", "pme_xpass_solve_kernel", K * K * pitch * (4 + 16), "x transform + solve + inverse x transform in one trip: theta 4 B + complex RW 16 B per point of the padded half-complex mesh")
# rocFFT's passes of the batched 2-D (y, z) transforms, in the bytes they move: the z pass is real <-> half-complex (K^3 x 4 B on the
# real side, K^2 x pitch x 8 B on the complex side; rocFFT runs it as a complex transform of length K / 2), the y pass a complex
# transform of length K in place on the half-complex mesh (K^2 x pitch x 8 B read and written)
for k in sorted(fk):
    m = re.search(r"len(\d+)", k)
    n = int(m.group(1)) if m else 0
    if n == K // 2: row(k[:34], k, pts * 4 + K * K * pitch * 8, "z pass, real <-> half-complex: K^3 x 4 B + K^2 x pitch x 8 B")
    elif n == K: row(k[:34], k, 2 * K * K * pitch * 8, "y pass, in place on the half-complex mesh: 2 x K^2 x pitch x 8 B")
    else: row(k[:34], k, 2 * K * K * pitch * 8, "a pass over the half-complex mesh: 2 x K^2 x pitch x 8 B")
row("water_step_kernel", "water_step_kernel", (sites // 4) * 496, "per rigid water: cluster + site records 80 B, posq / vel / force / ref rows of O, H, H (+ M) read once and written once: 496 B")
row("constrain_positions_kernel (SETTLE)", "constrain_positions_kernel", (sites // 4) * 3 * (32 + 32 + 16), "3 constrained atoms per water: pos RW, vel RW, ref R")
row("bonded_gather_kernel (Ewald excl.)", "bonded_gather_kernel", sites * (36 + 16 * 3), "36 B + 16 B x ~3 roles per site")
PY
tail -3 "$OUT/run.log"
find "$OUT" -name "*.db" -size +20M -delete

import os, subprocess, sys, math, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def run(arm):
    env = {k: v for k, v in os.environ.items() if not k.startswith("MDX_")}
    env["MDX_WATER_STEP"] = arm
    f = tempfile.mktemp(suffix=".npz")
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "water_step_child.py"), f], cwd=ROOT, env=env, check=True, capture_output=True)
    return np.load(f)
runs = [("0", run("0")), ("0", run("0")), ("1", run("1")), ("1", run("1"))]
for i in range(4):
    for j in range(i + 1, 4):
        out = []
        for name in ("tip3p_rigid", "opc", "opc_straddling_spme"):
            d = runs[i][1][name + "_pos"].astype(np.float64) - runs[j][1][name + "_pos"].astype(np.float64)
            d -= np.round(d / 24.8272) * 24.8272
            dv = runs[i][1][name + "_vel"].astype(np.float64) - runs[j][1][name + "_vel"].astype(np.float64)
            ea, eb = runs[i][1][name + "_e"], runs[j][1][name + "_e"]
            out.append(f"{name}: pos {math.sqrt((d ** 2).sum(1).mean()):.2e} vel {math.sqrt((dv ** 2).sum(1).mean()):.2e} dEpot {abs(ea[0] - eb[0]):.3f} dKE/KE {abs(ea[1] - eb[1]) / ea[1]:.1e} dvir {abs(ea[2] - eb[2]):.2f} of {abs(ea[2]):.0f}")
        print(f"arm {runs[i][0]} vs arm {runs[j][0]}: " + " | ".join(out))

"""Host-side bookkeeping of thermodynamic-integration windows, named after the `dynamics::alchemical` items
Molchanica imports (src/properties/water_sol.rs:19-21, 442, 516, 568): `LambdaWindow`, `collect_window`,
`free_energy_ti_with_sem`, `mean_coupled_interaction_kcal`.  Their bodies live in the absent crate; what is
built: sample mean and standard error of dH/dlambda per window (block averaging against correlation),
trapezoidal integration over lambda with the errors of the windows propagated in quadrature."""
from __future__ import annotations

import math
from dataclasses import dataclass, field


class AlchemicalError(ValueError):
    pass


@dataclass
class LambdaWindow:
    lam: float
    mean_dh_dl: float            # kcal/mol
    sem_dh_dl: float             # kcal/mol
    n_samples: int
    samples: list = field(default_factory=list, repr=False)

    @property
    def lambda_(self):           # `window.lambda` in the reference; `lambda` is a Python keyword
        return self.lam


def _sem(xs, n_blocks=5):
    """Standard error of the mean from block averages (falls back to the plain SEM for short series)."""
    n = len(xs)
    if n < 2:
        return 0.0
    if n < 2 * n_blocks:
        m = sum(xs) / n
        return math.sqrt(sum((x - m) ** 2 for x in xs) / (n - 1) / n)
    size = n // n_blocks
    means = [sum(xs[b * size:(b + 1) * size]) / size for b in range(n_blocks)]
    m = sum(means) / n_blocks
    return math.sqrt(sum((x - m) ** 2 for x in means) / (n_blocks - 1) / n_blocks)


def collect_window(lam: float, snapshots) -> LambdaWindow:
    """`collect_window(lambda, &md.snapshots)`: the snapshots' `dh_dlambda` samples of one window."""
    xs = [float(s["energy_data"]["dh_dlambda"] if "energy_data" in s else s["dh_dlambda"]) for s in snapshots]
    if not xs:
        raise AlchemicalError("no snapshots in the window")
    if any(not math.isfinite(x) for x in xs):
        raise AlchemicalError("non-finite dH/dlambda sample")
    return LambdaWindow(float(lam), sum(xs) / len(xs), _sem(xs), len(xs), xs)


def free_energy_ti_with_sem(windows) -> tuple[float, float]:
    """Trapezoidal integral of <dH/dlambda> over lambda and its standard error (kcal/mol)."""
    ws = sorted(windows, key=lambda w: w.lam)
    if len(ws) < 2:
        raise AlchemicalError("thermodynamic integration needs at least two windows")
    if any(b.lam <= a.lam for a, b in zip(ws, ws[1:])):
        raise AlchemicalError("duplicate lambda values")
    weights = []
    for k in range(len(ws)):
        lo = ws[k].lam - ws[k - 1].lam if k > 0 else 0.0
        hi = ws[k + 1].lam - ws[k].lam if k + 1 < len(ws) else 0.0
        weights.append(0.5 * (lo + hi))
    dg = sum(w * x.mean_dh_dl for w, x in zip(weights, ws))
    sem = math.sqrt(sum((w * x.sem_dh_dl) ** 2 for w, x in zip(weights, ws)))
    return dg, sem


def mean_coupled_interaction_kcal(snapshots):
    """Mean solute-environment interaction energy over the snapshots of a window (None without samples)."""
    xs = [float(s["energy_data"]["coupled_interaction"] if "energy_data" in s else s["coupled_interaction"]) for s in snapshots]
    return sum(xs) / len(xs) if xs else None

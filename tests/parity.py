"""Error metrics and bars of the parity tests (shared by every tests/test_*_gpu.py / test_hip_parity.py and by smoke()).

Three bars, because one absolute bar cannot serve quantities of different scale ("What's weak" 1):

* ``err``       returns / advantages / losses / logged statistics: ``max|a-b| / (1 + |b|)`` <= TOL = 1e-4 -- north_star's
                "returns/advantages/losses within 1e-4 fp32" (these quantities are O(1)).
* ``grad_err``  gradients: ``max|a-b| / max|b|`` <= GRAD_TOL = 1e-4 -- relative to the largest entry of the reference gradient
                (a typical entry of a golden's actor gradient is 1e-3; an absolute 1e-4 would let a 10 % error through).
* ``disp_err``  post-step parameters: the DISPLACEMENT of the optimiser step, ``max|(after-before) - (ref_after-before)| /
                max|ref_after-before|`` <= DISP_TOL = 1e-3 (an Adam step moves a parameter by ~lr = 8e-4; comparing the
                parameters themselves at 1e-4 would accept a step that is 12 % wrong).  Both sides start from the same `before`
                (the reference's parameters before that step), so the numerator is ``max|after - ref_after|``.

Every call records the observed value; tests/conftest.py writes the maxima per (metric, label) at session end
(profiles/r06_gputests.txt is a copy of that report from the GPU box).
"""
import numpy as np

TOL = 1e-4       # BASELINE.json north_star: returns / advantages / losses within 1e-4 fp32
GRAD_TOL = 1e-4  # gradients, relative to max|reference gradient|
DISP_TOL = 1e-3  # optimiser-step displacement, relative to max|reference displacement|

OBSERVED = {}    # (metric, label) -> [max observed, number of comparisons]


def _np(x):
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    return np.asarray(x, dtype=np.float64)


def _note(metric, label, v):
    o = OBSERVED.setdefault((metric, label), [0.0, 0])
    o[0] = max(o[0], v) if v == v else float("nan")
    o[1] += 1
    return v


def err(a, b, label=""):
    a, b = _np(a), _np(b)
    assert a.shape == b.shape or a.size == b.size, (a.shape, b.shape)
    return _note("err", label, float(np.max(np.abs(a.reshape(-1) - b.reshape(-1)) / (1.0 + np.abs(b.reshape(-1))))) if a.size else 0.0)


def grad_err(a, b, label=""):
    """max|a-b| relative to the largest entry of the reference gradient `b` (0 when both are all-zero)."""
    a, b = _np(a).reshape(-1), _np(b).reshape(-1)
    assert a.size == b.size, (a.size, b.size)
    if not a.size:
        return 0.0
    scale = float(np.max(np.abs(b)))
    d = float(np.max(np.abs(a - b)))
    if not np.isfinite(d):
        return _note("grad_err", label, float("inf"))
    return _note("grad_err", label, d / scale if scale > 0 else (0.0 if d == 0 else float("inf")))


def disp_err(after, ref_after, before, label=""):
    """Error of an optimiser step's displacement relative to the reference step's largest displacement; `before` = the reference's
    parameters before the step (what both sides started from)."""
    a, r, b = _np(after).reshape(-1), _np(ref_after).reshape(-1), _np(before).reshape(-1)
    assert a.size == r.size == b.size, (a.size, r.size, b.size)
    scale = float(np.max(np.abs(r - b)))
    d = float(np.max(np.abs(a - r)))
    if not np.isfinite(d):
        return _note("disp_err", label, float("inf"))
    return _note("disp_err", label, d / scale if scale > 0 else (0.0 if d == 0 else float("inf")))


def flat(params):
    """Flat fp64 vector of a parameter list in torch ``parameters()`` order."""
    return np.concatenate([_np(p).reshape(-1) for p in params]) if len(params) else np.zeros(0)


def golden_init(z, net):
    """Flat initial parameters of `net` ("actor" / "critic") from a golden .npz."""
    n = 0
    while f"{net}_init_{n}" in z.files:
        n += 1
    return flat([z[f"{net}_init_{i}"] for i in range(n)])


def golden_before(z, net, step):
    """The golden's parameters of `net` before optimiser step `step` (0-based): the initial weights, or the previous step's result."""
    return golden_init(z, net) if step == 0 else np.asarray(z[f"{net}_after"][step - 1], dtype=np.float64)


def check_grads(a, b, label="", tol=GRAD_TOL):
    v = grad_err(a, b, label)
    assert v <= tol, (label, v)


def check_step(after, ref_after, before, label="", tol=DISP_TOL):
    v = disp_err(after, ref_after, before, label)
    assert v <= tol, (label, v)


def report(path):
    lines = ["# observed maxima of the parity metrics (tests/parity.py): metric, label, max observed, comparisons, bar"]
    bars = {"err": TOL, "grad_err": GRAD_TOL, "disp_err": DISP_TOL}
    for (metric, label), (v, n) in sorted(OBSERVED.items()):
        lines.append(f"{metric:9s} {label or '-':40s} {v:.3e} {n:6d} {bars[metric]:.0e}")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")

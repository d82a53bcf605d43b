import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy view of one tests/golden/*.npz file; values come back as torch tensors."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __getitem__(self, key):
        return torch.from_numpy(self._z[key])

    def keys(self, prefix=""):
        return [k for k in self._z.files if k.startswith(prefix)]

    def state_dict(self, prefix):
        return {k[len(prefix):]: torch.from_numpy(self._z[k]) for k in self._z.files if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    """max|a - b| / max|b|: a true relative error, also for references far below 1 (parameter deltas of 1e-4, second
    order gradients of 1e-5).  An all-zero reference degenerates to the absolute error."""
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    scale = b.abs().max().item()
    return ((a - b).abs().max() / (scale if scale > 0 else 1.0)).item()


def check_step_trace(got, ref, tol_grad, tol_norm, tol_delta, lr_floor=0.05, history=None, resolution=None):
    """Compare one optimiser step, captured as {"grad.<name>", "gnorm", "delta.<name>"}, with the reference's.

    * pre-clip gradients: relative to max|ref| of each tensor (`tol_grad`), and the global norm (`tol_norm`);
    * parameter deltas p_after - p_before: Adam with beta1 = 0 moves an element by ~lr * sign(g) (first step) or
      lr * g / sqrt(v) -- where |g| is at rounding-noise level the sign is arbitrary, so deltas are compared only on
      elements whose reference gradient is above `lr_floor` of the tensor's largest, relative to max|ref delta|.
      Adam's second moment remembers EARLIER steps of the same parameter: an element whose earlier gradient carried a large
      RELATIVE error (a noise-level element of a step with large gradients) but still weighs in v = sum 0.001 g^2 inherits that
      error in its movement, whatever the current gradient.  ``history`` ({name: [(reference gradient, |got - reference|) of the
      parameter's earlier steps]}, appended to here) bounds what each element inherits -- v changes by sum 2 |g_p| |dg_p| over
      sum g^2, the movement by half of that -- and keeps an element out of the comparison when that bound exceeds half of
      `tol_delta`: there the movement says nothing about the optimiser's arithmetic, and the gradients themselves are held to
      `tol_grad` above.  Without this the check was sound only by luck: round 5 moved fp32 sums in their last bit and ONE
      element of one late step went from 3e-4 to 9e-3 on the plain-Adam path alone, with every gradient still within 3e-6.
      ``resolution`` ({name: |parameter| * 2^-23}): a movement is the difference of two fp32 parameter values, i.e. a whole
      number of units in the last place of the PARAMETER; two correct Adam implementations may round one unit apart, and when
      a step barely moves anything (the CutMix consistency step of the golden run: largest movement 1.3e-5 on weights of
      size 1) that one unit is 0.9 % of the largest movement.  One unit of the parameter's resolution is therefore allowed on
      top of `tol_delta` (round 5: that step's worst element was 2 ulp in the reference and 3 ulp here, gradients identical).
    Returns statistics: compared / total delta elements (callers assert a healthy share) and the worst errors."""
    names = sorted(k[len("grad."):] for k in ref if k.startswith("grad."))
    assert names and sorted(k[len("grad."):] for k in got if k.startswith("grad.")) == names
    bad = []
    compared = total = 0
    worst = {"grad": 0.0, "delta": 0.0}
    for n in names:
        g_ref, g_got = ref["grad." + n].double().cpu(), got["grad." + n].double().cpu()
        e = rel_err(g_got, g_ref)
        worst["grad"] = max(worst["grad"], e)
        if e > tol_grad:
            bad.append(("grad", n, e))
        d_ref, d_got = ref["delta." + n].double().cpu(), got["delta." + n].double().cpu()
        gmax, dmax = g_ref.abs().max().item(), d_ref.abs().max().item()
        if gmax == 0.0:          # a parameter this step's graph does not reach: zero gradient, Adam must not move it
            if g_got.abs().max().item() != 0.0 or d_got.abs().max().item() != 0.0:
                bad.append(("moved-without-gradient", n, d_got.abs().max().item()))
            continue
        if dmax == 0.0:
            continue
        mask = g_ref.abs() > lr_floor * gmax
        if history is not None and history.get(n):
            num = sum(2.0 * gp.abs() * ep for gp, ep in history[n])
            den = sum(gp.square() for gp, _ in history[n]) + g_ref.square()
            mask &= 0.5 * num <= 0.5 * tol_delta * den
        total += mask.numel()
        compared += int(mask.sum())
        d_err = (d_got - d_ref).abs()
        if resolution is not None and n in resolution:
            d_err = (d_err - resolution[n].double().cpu().reshape(d_err.shape)).clamp_min(0.0)
        e = (d_err * mask).max().item() / dmax
        worst["delta"] = max(worst["delta"], e)
        if e > tol_delta:
            bad.append(("delta", n, e))
    e = abs(float(got["gnorm"]) - float(ref["gnorm"])) / max(abs(float(ref["gnorm"])), 1e-30)
    if e > tol_norm:
        bad.append(("gnorm", "", e))
    if history is not None:
        for n in names:
            g_ref = ref["grad." + n].double().cpu()
            history.setdefault(n, []).append((g_ref, (got["grad." + n].double().cpu() - g_ref).abs()))
    assert not bad, bad[:12]
    return {"compared": compared, "total": total, "worst_grad": worst["grad"], "worst_delta": worst["delta"],
            "gnorm_err": e}

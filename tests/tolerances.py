"""Tolerances of the parity tests, in one place.

Every gradient check is RELATIVE TO THE TENSOR'S OWN SCALE: max|got - ref| <= rel * max|ref|.  There is no absolute floor
(round 2 used max(1, max|ref|), which made the checks of small-magnitude tensors -- every gradient of an mse='mean' step is below
0.06 -- nearly vacuous); the only absolute term is 1e-7 of the largest gradient of the whole step (`step_scale`), i.e. fp32 rounding
of the upstream gradients a tensor's entries are summed from, which matters only for tensors that are exactly or almost zero.
Measured errors of the HIP path against the reference's fixtures: 0.5-1.1e-6 of the tensor's max for every tensor of every fixture.
"""
import numpy as np

TINY_REL = 2e-5         # fixtures computed by the reference itself at the same sizes (H <= 64, B <= 40)
BIG_REL = 3e-4          # batch / hidden sizes where the fp32 reference's own summation order moves the last digits
LOSS_REL = 1e-4


def assert_grad_close(got, ref, rel, name="", step_scale=0.0):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = float(np.abs(ref).max()) if ref.size else 0.0
    err = float(np.abs(got - ref).max()) if ref.size else 0.0
    tol = rel * scale + 1e-7 * step_scale
    assert err <= tol, f"{name}: max|err| {err:.3e} > {tol:.3e} = {rel:g} x max|ref| {scale:.3e} (+1e-7 x step scale {step_scale:.3e}); relative {err / max(scale, 1e-300):.2e}"
    return err / scale if scale > 0 else 0.0


def step_scale_of(ref_grads):
    """Largest |gradient| over all tensors of one step: the scale of the upstream gradients."""
    return max((float(np.abs(np.asarray(r)).max()) for r in ref_grads if np.asarray(r).size), default=0.0)


def assert_loss_close(got, ref, rel=LOSS_REL, name=""):
    got, ref = float(got), float(ref)
    assert abs(got - ref) <= rel * abs(ref) + 1e-9, f"{name}: {got!r} vs {ref!r} (rel {abs(got - ref) / max(abs(ref), 1e-300):.2e} > {rel:g})"

"""Shared comparison helpers of the parity tests.

north_star's bar is "1e-3 relative fp32 PER-PIXEL" for outputs; a max-norm check lets small-magnitude pixels be off
by orders of magnitude.  `per_elem_err` is the per-element relative error with a floor of `floor`*max|ref| on the
denominator (so that exact zeros do not demand exact equality):
        err_i = |a_i - b_i| / max(|b_i|, floor * max|b|)
`assert_close_per_elem` asserts max_i err_i <= tol.
"""
import torch


def per_elem_err(a, b, floor=1e-3):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    mx = b.abs().max().clamp_min(1e-300)
    return (a - b).abs() / torch.maximum(b.abs(), floor * mx)


def assert_close_per_elem(a, b, tol=1e-3, floor=1e-3, what=""):
    assert tuple(a.shape) == tuple(b.shape), (what, tuple(a.shape), tuple(b.shape))
    e = per_elem_err(a, b, floor)
    worst = e.max().item()
    assert worst <= tol, "%s: per-element relative error %.3e > %.1e (%d of %d elements above tol; ref max %.3e)" % (
        what, worst, tol, int((e > tol).sum()), e.numel(), b.abs().max().item())
    return worst


def max_rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()

"""Helpers shared by the tests."""
import torch


def report(name, got, want, rtol, atol):
    """Assert closeness with a diagnostic that survives a remote run: worst element, its location, and the
    share of elements out of tolerance."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    assert got.shape == want.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(want.shape)}"
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    if bad.any() or not torch.isfinite(got).all():
        idx = torch.nonzero(bad)
        worst = int(torch.argmax(err - tol))
        wi = tuple(int(v) for v in torch.unravel_index(torch.tensor(worst), err.shape))
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{bad.numel()} out of tolerance (rtol={rtol}, atol={atol}); worst at {wi}: got "
            f"{got[wi].item():.6g} want {want[wi].item():.6g}; first bad idx {idx[0].tolist() if len(idx) else None}; "
            f"max|want|={want.abs().max().item():.4g} finite={bool(torch.isfinite(got).all())}")


def rel_l2(got, want):
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).norm() / (want.norm() + 1e-30))

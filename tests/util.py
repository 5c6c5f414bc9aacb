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


# ---------------------------------------------------------------------------------------------------------------
# The InstanceNorm U-Net variants of models/unetvgg.py (fixtures in tests/golden/variants.npz): case name ->
# (product class name, product kwargs, oracle class name, oracle kwargs); all take one input channel.
# ---------------------------------------------------------------------------------------------------------------
VARIANTS = {
    "unetsimple_c4": ("UNetSimple", dict(out_classes=4, in_classes=1), "RefUNetSimple", dict(out_classes=4)),
    "unetsimple_cs32_c3": ("UNetSimple", dict(out_classes=3, in_classes=1, channelscale=32), "RefUNetSimple", dict(out_classes=3, channelscale=32)),
    "unetsimple_flat_cs16_c4": ("UNetSimple", dict(out_classes=4, in_classes=1, channelscale=16, noupdownsam=True),
                                "RefUNetSimple", dict(out_classes=4, channelscale=16, noupdownsam=True)),
    "unetnaive_c4": ("UNetNaive", dict(out_classes=4, in_classes=1), "RefUNetNaive", dict(out_classes=4)),
    "unetnaive_convtrans_pool_c3": ("UNetNaive", dict(out_classes=3, in_classes=1, if_convtrans=True, ifmaxpool=True),
                                    "RefUNetNaive", dict(out_classes=3, if_convtrans=True, ifmaxpool=True)),
    "unetnocenter_c4": ("UNetWithoutCenter", dict(out_classes=4, in_classes=1), "RefUNetNaive", dict(out_classes=4, center=False)),
    # networks that also return feature maps (their loss adds FEATURE_WEIGHT * mean(f^2) per map, see step_loss)
    "unetsimple_multi_cs32_c3": ("UNetSimple", dict(out_classes=3, in_classes=1, channelscale=32, multiout=True),
                                 "RefUNetSimple", dict(out_classes=3, channelscale=32, multiout=True)),
    "unetnaivemulti_c4": ("UNetNaiveMultiOut", dict(out_classes=4, in_classes=1), "RefUNetFourLevel", dict(out_classes=4)),
    "unetlimmulti_c4": ("UNetLimMultiOut", dict(out_classes=4, in_classes=1, multiout=True), "RefUNetFourLevel", dict(out_classes=4, channel_depth=16)),
    "spidernet_c4": ("SpiderNet", dict(out_classes=4, in_classes=1), "RefSpiderNet", dict(out_classes=4)),
    "spidernet_multi_cs32_c3": ("SpiderNet", dict(out_classes=3, in_classes=1, channelscale=32, multiout=True),
                                "RefSpiderNet", dict(out_classes=3, channelscale=32, multiout=True)),
    "unettetra_cs32_c4": ("UNetTetra", dict(out_classes=4, in_classes=1, channelscale=32), "RefUNetTetra", dict(out_classes=4, channelscale=32)),
    "unetskip_multi_c4": ("UNetSkip", dict(out_classes=4, in_classes=1, multiout=True), "RefUNetSimple", dict(out_classes=4, channelscale=32, multiout="dec4")),
    "unetsimplereduced_cs32_c3": ("UNetSimpleReduced", dict(out_classes=3, in_classes=1, channelscale=32), "RefUNetSimpleReduced", dict(out_classes=3, channelscale=32)),
}
FEATURE_WEIGHT = 0.05


def split_outputs(out):
    """model output -> (logits, list of extra feature maps)."""
    if not isinstance(out, tuple):
        return out, []
    return out[0], list(out[1]) if isinstance(out[1], (list, tuple)) else [out[1]]


def step_loss(out, t, dice, focal):
    """The loss the fixtures were generated with: Dice + focal on the logits, plus FEATURE_WEIGHT * mean(f^2) per extra map."""
    logits, feats = split_outputs(out)
    loss = dice(logits, t) + focal(logits, t)
    for f in feats:
        loss = loss + FEATURE_WEIGHT * (f * f).mean()
    return logits, feats, loss


def variant_product(name):
    import models.unetvgg as M
    cls, kw, _, _ = VARIANTS[name]
    return getattr(M, cls)(**kw)


def variant_oracle(name):
    from oracle import ref_models as RM
    _, _, cls, kw = VARIANTS[name]
    return getattr(RM, cls)(**kw)


def input_channels(name):
    return 1 if (name in VARIANTS or "inout" in name or "nested" in name) else 3

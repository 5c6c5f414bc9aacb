"""A well-conditioned UNetVgg case for end-to-end GRADIENT parity (test infrastructure, like the rest of oracle/).

On a random-weight network the fp32 gradients of ANY implementation, the reference's included, sit 1e-3..1e-2 from exact
arithmetic: a handful of pre-ReLU values lie within rounding of zero, their sign (the ReLU mask) differs between two
roundings, and each such flip moves the gradient by ~1e-3.  `condition_` removes almost all of those sites: BatchNorm
beta in [3, 4] with gamma in [0.4, 0.6] and convolution biases in [4, 5] keep every BatchNorm / bias ReLU far inside its
linear range, so only the InstanceNorm -> up-sample -> ReLU sites of the decoder (no affine, values cross zero) remain,
and the batch tag is chosen so that the smallest |value| among them is well above fp32 rounding."""
import torch

from . import ref_models as RM

TAGS = ("wellcond:8", "wellcond:7")      # chosen among twelve candidates: the two with no value within rounding of a ReLU threshold


def condition_(m):
    sd = m.state_dict()
    for k, t in sd.items():
        if t.dim() != 1:
            continue
        is_bn = (k.rsplit(".", 1)[0] + ".running_mean") in sd
        if k.endswith("bias") and is_bn:
            t.copy_(3.0 + RM.hash_uniform("wc:" + k, t.shape, 0, 1))
        elif k.endswith("bias") and not k.startswith("final"):
            t.copy_(4.0 + RM.hash_uniform("wc:" + k, t.shape, 0, 1))
        elif k.endswith("weight") and is_bn:
            t.copy_(RM.hash_uniform("wc:" + k, t.shape, 0.4, 0.6))
    m.load_state_dict(sd)
    return m


def batch(tag=TAGS[0]):
    return RM.synthetic_batch(2, 64, 4, in_ch=3, tag=tag)


def relu_margin(model, x):
    """(smallest |pre-ReLU value| relative to the mean magnitude of its tensor, number of values below 1e-5 of that
    mean) over every ReLU of the network, from one forward pass."""
    seen = []
    hooks = [mod.register_forward_pre_hook(lambda mod, inp: seen.append(inp[0].detach().abs()))
             for mod in model.modules() if isinstance(mod, torch.nn.ReLU)]
    with torch.no_grad():
        model(x)
    for h in hooks:
        h.remove()
    rel = [(a / a.mean()) for a in seen]
    return min(float(r.min()) for r in rel), sum(int((r < 1e-5).sum()) for r in rel)

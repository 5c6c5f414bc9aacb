"""CPU oracle for the pixel-mask losses.  TEST INFRASTRUCTURE ONLY.

Pure-torch restatement of the arithmetic of the reference's losses, written from the formulas the
reference code evaluates (not from its text):

* soft squared Dice        -- /root/reference/loss/dice.py:25-87  (loss = 1 - coefficient, :22-23)
* self-adjusting Dice, log-cosh Dice -- /root/reference/loss/dice.py:275-349
* arg-max Dice metric      -- /root/reference/loss/dice.py:89-184
* multi-class focal loss   -- /root/reference/loss/focal.py:32-75, 108-137
* edge-aware Dice          -- /root/reference/loss/hausdorff.py:71-128, /root/reference/loss/edge_loss.py:32-55

Pinned by `oracle/make_golden.py`, which imports the reference itself in the build container and
checks these functions against it bit for bit; the recorded outputs live in tests/golden/.
Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this module.
"""
import torch
import torch.nn.functional as F
from torch import nn


def _one_hot(labels, n_classes, dtype=torch.float32):
    """[N,H,W] integer labels -> [N,C,H,W] one-hot; H must equal W (dice.py:48 views with shape[2] twice)."""
    n, side = labels.shape[0], labels.shape[2]
    idx = labels.view(n, 1, side, side)
    return torch.zeros(n, n_classes, side, side, dtype=dtype, device=labels.device).scatter(1, idx, 1)


def _fg_images(act):
    """Indices of batch items that contain any non-zero label (dice.py:82)."""
    return torch.unique(torch.nonzero(act, as_tuple=True)[0])


def _average_over_fg(dice, act, all_channels, guard_single_channel):
    """Drop channel 0 unless asked, then average over images that have foreground (dice.py:74-86)."""
    n_ch = dice.shape[1]
    kept = dice if (all_channels or n_ch == 1) else dice[:, 1:]
    per_img = kept.shape[1]
    div = kept.numel()
    sel = _fg_images(act)
    if len(sel) and (n_ch > 1 or not guard_single_channel):
        div = len(sel) * per_img
    return kept[sel].sum() / div, kept


def dice_coefficient_sq(probs, act, ignore_onlybg=False, no_sft=False, is_all_chann=False):
    hot = _one_hot(act, probs.shape[1]) if act.dim() != 4 else act
    p = probs if no_sft else torch.softmax(probs, dim=1)
    inter = (p * hot).flatten(2).sum(2)
    psq = (p * p).flatten(2).sum(2)
    gsq = (hot * hot).flatten(2).sum(2)
    dice = (2 * inter + 1) / (psq + gsq + 1)
    coeff, _ = _average_over_fg(dice, act, is_all_chann, guard_single_channel=True)
    return coeff


def dice_loss_sq(pred, act, no_sft=False, is_all_chann=False):
    return 1 - dice_coefficient_sq(pred, act, no_sft=no_sft, is_all_chann=is_all_chann)


def dice_coefficient_sa(probs, act, ignore_onlybg=False, no_sft=False):
    """Self-adjusting Dice (dice.py:275-325): both the overlap and the prediction mass are weighted by 1 - p."""
    hot = _one_hot(act, probs.shape[1]) if act.dim() != 4 else act
    p = probs if no_sft else torch.softmax(probs, dim=1)
    wgt = 1 - p
    inter = (p * hot * wgt).flatten(2).sum(2)
    pmass = (wgt * p).flatten(2).sum(2)
    gmass = hot.clone().flatten(2).sum(2)
    dice = (2 * inter + 1) / (pmass + gmass + 1)
    coeff, _ = _average_over_fg(dice, act, False, guard_single_channel=True)
    return coeff


def dice_loss_sa(pred, act, no_sft=False):
    return 1 - dice_coefficient_sa(pred, act, no_sft=no_sft)


def logcoshdiceloss(pred, act, no_sft=False):
    """dice.py:347-349"""
    return torch.log(torch.cosh(dice_loss_sq(pred, act, no_sft=no_sft)))


def dice_coefficient(probs, act, is_list_bat=False, nosft=False, channelcnt=None, is_all_chann=False):
    if act.dim() != 4:
        if not channelcnt:
            channelcnt = probs.shape[-3]
        hot = _one_hot(act, channelcnt)
    else:
        hot = act
    p = probs if nosft else torch.softmax(probs, dim=1)
    if p.dim() != 3:
        p = torch.argmax(p, dim=1)
    p = p.unsqueeze(1)
    phot = torch.zeros(p.shape[0], channelcnt, *p.shape[-2:], device=p.device).scatter(1, p.long(), 1)
    inter = (phot * hot).flatten(2).sum(2)
    dice = (2 * inter + 1) / (phot.flatten(2).sum(2) + hot.flatten(2).sum(2) + 1)
    coeff, kept = _average_over_fg(dice, act, is_all_chann, guard_single_channel=False)
    return kept if is_list_bat else coeff


def focal_loss_with_logits(p, t, gamma=2., alpha=None, reduction='elementwise_mean'):
    ce = F.cross_entropy(p, t, weight=alpha, reduction='none')
    q = F.softmax(p, dim=1)
    hot = torch.zeros(t.shape[0], p.shape[1], *t.shape[1:], dtype=t.dtype, device=t.device).scatter_(1, t.unsqueeze(1), 1)
    w = 1 - torch.where(hot == 1., q, 1 - q)
    w.pow_(gamma)
    out = w * 1. * ce.unsqueeze(1)
    if reduction == 'elementwise_mean':
        return out.mean()
    if reduction == 'none':
        return out
    if reduction == 'sum':
        return out.sum()
    raise ValueError('Reduction parameter unknown.')



def dice_coefficientold(probs, act, is_list_bat=False, nosft=False, channelcnt=None):
    """/root/reference/loss/dice.py:186-256: soft Dice with un-squared denominators."""
    act = act.view(act.shape[0], 1, act.shape[2], act.shape[2])
    hot = torch.zeros(act.shape[0], channelcnt if channelcnt else probs.shape[1], *act.shape[2:]).scatter(1, act, 1)
    if probs.shape[-3] == 1 and probs.dim() == 4 or probs.dim() == 3:
        if not channelcnt:
            channelcnt = act.max() + 1
        probs = torch.zeros(probs.shape[0], channelcnt, *probs.shape[2:]).scatter(1, probs.long(), 1)
    elif not nosft:
        probs = nn.Softmax2d()(probs)

    def plane_sum(v):
        return torch.sum(v.view(v.shape[0], v.shape[1], -1), dim=2)
    dice = (2 * plane_sum(probs * hot) + 1) / (plane_sum(probs) + plane_sum(hot) + 1)
    fg = dice[:, 1:]
    if is_list_bat:
        return fg
    div_by = torch.numel(fg)
    rows = torch.unique(torch.nonzero(act, as_tuple=True)[0])
    if len(rows):
        div_by = len(rows) * torch.numel(dice[0, 1:])
    return torch.sum(fg[rows]) / div_by


def focal_from_probabilities(p, t, gamma, loss_val, reduction):
    """/root/reference/loss/focal.py:32-106 (_focal_loss -> _general_focal_loss with class weight 1)."""
    onehot = torch.zeros_like(p).scatter_(1, t.unsqueeze(1), 1)
    weight = 1 - torch.where(torch.eq(onehot, 1.), p, 1 - p)
    weight.pow_(gamma)
    if loss_val.dim() < weight.dim():
        loss_val = loss_val.unsqueeze(1)
    val = weight * 1. * loss_val
    return {"elementwise_mean": torch.mean, "sum": torch.sum, "none": lambda v: v}[reduction](val)


# ---------------------------------------------------------------------------------------------
# Edge-aware Dice (/root/reference/loss/hausdorff.py:71-128, /root/reference/loss/edge_loss.py:32-55)
# ---------------------------------------------------------------------------------------------
def get_hot_enc(labels, channels=3):
    if labels.dim() == 2:
        labels = labels[None, None]
    if labels.dim() == 3:
        labels = labels.view(labels.shape[0], 1, labels.shape[2], labels.shape[2])
    return torch.zeros(labels.shape[0], channels, *labels.shape[2:], device=labels.device).scatter(1, labels.long(), 1)


def get_edg_c(tens):
    """Four depthwise Sobel responses (x, -x, y, -y), each InstanceNorm2d'ed and rectified, summed in the reference's order."""
    if tens.dim() == 3:
        tens = tens.float().unsqueeze(1)
    C = tens.shape[1]
    gx = torch.tensor([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=torch.float32)
    gy = torch.tensor([[1, 2, 1], [0, 0, 0], [-1, -2, -1]], dtype=torch.float32)
    norm, out = nn.InstanceNorm2d(C), None
    for k in (gx, -gx, gy, -gy):
        w = k.unsqueeze(0).expand(C, 1, 3, 3).to(tens.device)
        r = torch.relu(norm(F.conv2d(tens, w, stride=1, padding=1, groups=C)))
        out = r if out is None else out + r
    return out


def edge_conv_loss(probs, act):
    return dice_loss_sq(get_edg_c(probs), get_edg_c(get_hot_enc(act)))


def dice_focal_closed_form_grad(logits, labels, gamma=2.0, w_dice=1.0, w_focal=1.0):
    """d(w_dice*dice_loss_sq + w_focal*focal)/dlogits from the closed forms the HIP backward kernel
    implements; double precision, used to cross-check autograd and the kernel."""
    z = logits.double()
    n, c, h, w = z.shape
    p = torch.softmax(z, 1)
    g = _one_hot(labels, c, dtype=torch.float64)
    A = (p * g).flatten(2).sum(2)
    B = (p * p).flatten(2).sum(2)
    G = g.flatten(2).sum(2)
    den = B + G + 1
    sel = (labels.flatten(1) != 0).any(1)
    nsel = int(sel.sum())
    u = torch.zeros_like(p)
    if nsel and c > 1:
        div = nsel * (c - 1)
        coef = -(2 * g * den[:, :, None, None] - 2 * p * (2 * A + 1)[:, :, None, None]) / (den * den)[:, :, None, None] / div
        coef[:, 0] = 0
        coef[~sel] = 0
        u = coef
    gd = p * (u - (u * p).sum(1, keepdim=True))
    logp = torch.log_softmax(z, 1)
    ce = -(logp * g).sum(1, keepdim=True)
    pt = (p * g).sum(1, keepdim=True)
    S = (1 - pt) ** gamma + ((p ** gamma) * (1 - g)).sum(1, keepdim=True)
    if gamma == 0:
        v = -g * S / pt
    else:
        v = torch.where(g == 1, -S / pt - ce * gamma * (1 - pt) ** (gamma - 1), ce * gamma * p ** (gamma - 1))
    gf = p * (v - (v * p).sum(1, keepdim=True)) / (n * c * h * w)
    return w_dice * gd + w_focal * gf

"""Fused softmax + squared-Dice + focal loss on the GPU (autograd.Function over the C ABI) and the
arg-max Dice metric.  CPU tensors are served by the same formulas in torch ops (`_host_*`)."""
import threading
import weakref

import torch
import torch.nn.functional as F

from . import _lib, ops
from .ops import dt, workspace
from ._lib import call, query

NO_SFT, ALL_CHANN, FOCAL_SUM, SELF_ADJ = 1, 2, 4, 8


def _env(t):
    dev = t.device.index if t.device.index is not None else torch.cuda.current_device()
    return dev, torch.cuda.current_stream(dev).cuda_stream


class _DiceFocalFn(torch.autograd.Function):
    """(logits, target) -> (dice_coefficient, focal_loss); one forward kernel computes the per-(n,c) sums both
    losses need, one backward kernel writes dlogits for any mix of the two upstream gradients."""

    @staticmethod
    def forward(ctx, logits, target, gamma, alpha, flags, want_map):
        _lib.lib()
        z = logits.contiguous()
        if z.dtype not in (torch.float32, torch.bfloat16):
            z = z.float()
        N, C = z.shape[0], z.shape[1]
        HW = z[0, 0].numel()
        labels = hot = None
        if target.dtype == torch.int64:
            labels = target.contiguous()
        else:
            hot = target.contiguous().float()
        al = None if alpha is None else alpha.to(device=z.device, dtype=torch.float32).contiguous()
        dev, st = _env(z)
        sums = torch.empty(N * C * 4, dtype=torch.float32, device=z.device)
        iflags = torch.empty(N, dtype=torch.int32, device=z.device)
        scalars = torch.empty(4, dtype=torch.float32, device=z.device)
        fmap = torch.empty(z.shape, dtype=torch.float32, device=z.device) if want_map else None
        ws = workspace(query("segk_dice_focal_workspace", N, C, HW), z.device)
        p = lambda t: None if t is None else t.data_ptr()
        tbytes = 8 if labels is not None else 4 * C
        with ops._streamed("dice + focal loss", N * HW * (C * z.element_size() + tbytes)):
            call("segk_dice_focal_fwd", z.data_ptr(), dt(z.dtype), p(labels), p(hot), N, C, HW, float(gamma), p(al), flags,
                 sums.data_ptr(), iflags.data_ptr(), p(fmap), ws.data_ptr(), ws.numel(), dev, st)
            call("segk_dice_focal_finalize", sums.data_ptr(), iflags.data_ptr(), N, C, HW, flags, scalars.data_ptr(), dev, st)
        ctx.save_for_backward(z, labels if labels is not None else hot, sums, iflags, scalars)
        ctx.meta = (N, C, HW, float(gamma), al, flags, labels is not None, logits.dtype)
        ctx.token = _store().get("pending")      # set by dice_focal(): marks the shared node as spent once it has run backward
        ctx.set_materialize_grads(False)
        outs = (scalars[0], scalars[1])
        if want_map:
            ctx.mark_non_differentiable(fmap)
            return outs + (fmap,)
        return outs

    @staticmethod
    def backward(ctx, g_dice, g_focal, *unused):
        z, tgt, sums, iflags, scalars = ctx.saved_tensors
        N, C, HW, gamma, al, flags, is_labels, in_dtype = ctx.meta
        if ctx.token is not None:
            ctx.token["spent"] = True
        gs = torch.zeros(2, dtype=torch.float32, device=z.device)
        if g_dice is not None:
            gs[0] = -g_dice.float()          # the kernel differentiates the loss 1 - coefficient
        if g_focal is not None:
            gs[1] = g_focal.float()
        dz = torch.empty_like(z)
        dev, st = _env(z)
        p = lambda t: None if t is None else t.data_ptr()
        with ops._streamed("dice + focal loss", N * HW * (2 * C * z.element_size() + (8 if is_labels else 4 * C))):
            call("segk_dice_focal_bwd", z.data_ptr(), dt(z.dtype), p(tgt) if is_labels else None, None if is_labels else p(tgt),
                 N, C, HW, gamma, p(al), flags, sums.data_ptr(), iflags.data_ptr(), scalars.data_ptr(), gs.data_ptr(),
                 dz.data_ptr(), dev, st)
        return dz.to(in_dtype), None, None, None, None, None


def _as_square_labels(act):
    # the reference reshapes with shape[2] twice (loss/dice.py:48): non-square targets raise exactly there
    return act.view(act.shape[0], 1, act.shape[2], act.shape[2])


# The training loops add the two losses of ONE (logits, target) pair: `dice_loss_sq(out, t) + focal_loss_with_logits(out, t)`
# (perlosstrain/naive_train.py:421-423, pl_utils/pl_mod.py:64-67).  The fused kernels compute both, so the second call
# re-uses the autograd node of the first: one forward launch, and one backward launch that receives both upstream
# gradients and writes dlogits once.  The entry is valid while both tensors are alive and unmodified, the focal
# parameters agree, and the node has not run its backward yet.  It is kept per host thread (two training loops in one
# process do not see each other's entry), holds the logits weakly and is dropped as soon as the logits tensor dies, the
# second loss has been served, or the node has run its backward; until then it keeps the target alive (so that its address
# cannot be handed to another tensor and matched by mistake) together with the two 0-dim results.
_SHARED = threading.local()


def _same_data(a, b):
    """Two tensor objects over the same elements (the Dice entry point hands the kernels a reshaped view of the labels)."""
    return (a is b) or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride() and a.dtype == b.dtype
                        and a._version == b._version)


def _alpha_key(alpha):
    """The class weights by value: a list object's id can be re-used by another list, and a tensor edited in place keeps its id."""
    if alpha is None:
        return None
    if isinstance(alpha, torch.Tensor):
        return ("tensor", alpha.data_ptr(), alpha._version, tuple(alpha.shape), alpha.dtype)
    return tuple(float(a) for a in alpha)


def _store():
    """This host thread's slot dictionary (captured by the weak-reference callback, which may run on another thread)."""
    return _SHARED.__dict__


def _shared_lookup(pred, target, gamma, alpha, flags, grad):
    store = _store()
    e = store.get("entry")
    if e is None:
        return None
    if e["token"]["spent"]:
        store["entry"] = None
        return None
    if e["pred"]() is not pred or e["versions"] != (pred._version, target._version) or not _same_data(e["target"], target):
        return None
    if e["key"] != (float(gamma), _alpha_key(alpha), flags, grad):
        return None
    store["entry"] = None                     # the second loss of the pair has been served
    return e["outs"]


def dice_focal(pred, target, gamma=2.0, alpha=None, no_sft=False, is_all_chann=False, focal_sum=False, want_map=False,
               self_adjusting=False):
    flags = (NO_SFT if no_sft else 0) | (ALL_CHANN if is_all_chann else 0) | (FOCAL_SUM if focal_sum else 0) | \
        (SELF_ADJ if self_adjusting else 0)
    if want_map:
        return _DiceFocalFn.apply(pred, target, gamma, alpha, flags, True)
    grad = torch.is_grad_enabled() and pred.requires_grad
    outs = _shared_lookup(pred, target, gamma, alpha, flags, grad)
    if outs is not None:
        return outs
    token = {"spent": False}
    store = _store()
    store["pending"] = token
    try:
        outs = _DiceFocalFn.apply(pred, target, gamma, alpha, flags, False)
    finally:
        store["pending"] = None

    def gone(_ref, token=token, store=store):        # the logits tensor died: nothing can match this entry any more
        e = store.get("entry")
        if e is not None and e["token"] is token:
            store["entry"] = None
    store["entry"] = {"pred": weakref.ref(pred, gone), "target": target, "versions": (pred._version, target._version),
                     "key": (float(gamma), _alpha_key(alpha), flags, grad), "token": token, "outs": outs}
    return outs


# ------------------------------------------------------------------------------------ public: Dice
def dice_coefficient_sq(probs, act, ignore_onlybg=False, no_sft=False, is_all_chann=False, _self_adjusting=False):
    if not probs.is_cuda:
        return _host_dice_sq(probs, act, no_sft, is_all_chann, _self_adjusting)
    if act.dim() != 4:
        sq = _as_square_labels(act)
        if sq.dtype != torch.int64:
            raise RuntimeError("scatter(): Expected dtype int64 for index")
        target = sq.view(sq.shape[0], sq.shape[2], sq.shape[3])
    else:
        target = act.float()
    coeff, _ = dice_focal(probs, target, no_sft=no_sft, is_all_chann=is_all_chann, self_adjusting=_self_adjusting)
    return coeff


def dice_loss_sq(pred, act, no_sft=False, is_all_chann=False):
    return 1 - dice_coefficient_sq(pred, act, no_sft=no_sft, is_all_chann=is_all_chann)


def dice_coefficient_sa(probs, act, ignore_onlybg=False, no_sft=False):
    """Self-adjusting Dice coefficient (reference loss/dice.py:275-325): (2 sum p g (1-p) + 1) / (sum p (1-p) + sum g + 1),
    background channel dropped, averaged over the images that contain foreground."""
    return dice_coefficient_sq(probs, act, no_sft=no_sft, _self_adjusting=True)


def dice_loss_sa(pred, act, no_sft=False):
    """reference loss/dice.py:327-345"""
    return 1 - dice_coefficient_sa(pred, act, no_sft=no_sft)


def logcoshdiceloss(pred, act, no_sft=False):
    """log(cosh(squared-Dice loss)) (reference loss/dice.py:347-349); the scalar chain rule is autograd's."""
    return torch.log(torch.cosh(dice_loss_sq(pred, act, no_sft=no_sft)))


def dice_coefficient(probs, act, is_list_bat=False, nosft=False, channelcnt=None, is_all_chann=False):
    """Arg-max Dice (validation metric, not differentiable)."""
    if not probs.is_cuda or act.dim() == 4 or probs.dim() != 4 or (channelcnt and channelcnt != probs.shape[1]):
        if probs.is_cuda and (act.dim() == 4 or probs.dim() != 4 or channelcnt):
            raise NotImplementedError("the HIP metric kernel takes [N,C,H,W] scores and [N,H,W] integer labels")
        return _host_dice_metric(probs, act, is_list_bat, nosft, channelcnt, is_all_chann)
    _lib.lib()
    sq = _as_square_labels(act)
    if sq.dtype != torch.int64:
        raise RuntimeError("scatter(): Expected dtype int64 for index")
    z = probs.detach().contiguous()
    if z.dtype not in (torch.float32, torch.bfloat16):
        z = z.float()
    N, C = z.shape[0], z.shape[1]
    HW = z[0, 0].numel()
    counts = torch.empty(N, C, 3, dtype=torch.float32, device=z.device)
    iflags = torch.empty(N, dtype=torch.int32, device=z.device)
    dev, st = _env(z)
    call("segk_argmax_dice", z.data_ptr(), dt(z.dtype), sq.contiguous().data_ptr(), N, C, HW, None, counts.data_ptr(),
         iflags.data_ptr(), dev, st)
    dice = (2 * counts[..., 0] + 1) / (counts[..., 1] + counts[..., 2] + 1)
    kept = dice if (is_all_chann or C == 1) else dice[:, 1:]
    if is_list_bat:
        return kept
    sel = iflags.bool()
    nsel = sel.sum()
    div = torch.where(nsel > 0, nsel * kept.shape[1], torch.tensor(kept.numel(), device=z.device))
    return (kept * sel[:, None]).sum() / div


def _dice_sums(z, labels, flags):
    """Per-(image, class) sums of the fused forward kernel: [N, C, 4] = (sum p g, sum p^2, sum g, focal) -- with SELF_ADJ
    (sum p g (1-p), sum p (1-p), sum g, -) -- and the per-image 'has foreground' flags."""
    N, C = z.shape[0], z.shape[1]
    HW = z[0, 0].numel()
    dev, st = _env(z)
    sums = torch.empty(N * C * 4, dtype=torch.float32, device=z.device)
    iflags = torch.empty(N, dtype=torch.int32, device=z.device)
    ws = workspace(query("segk_dice_focal_workspace", N, C, HW), z.device)
    call("segk_dice_focal_fwd", z.data_ptr(), dt(z.dtype), labels.data_ptr(), None, N, C, HW, 2.0, None, flags,
         sums.data_ptr(), iflags.data_ptr(), None, ws.data_ptr(), ws.numel(), dev, st)
    return sums.view(N, C, 4), iflags


def dice_coefficientold(probs, act, is_list_bat=False, nosft=False, channelcnt=None):
    """The earlier, un-squared soft Dice (reference loss/dice.py:186-256): (2 sum p g + 1) / (sum p + sum g + 1), background
    dropped, averaged over the images with foreground.  On CUDA scores it is a metric (no gradient): two passes of the fused
    forward kernel give sum p = sum p^2 + sum p (1 - p)."""
    plain = probs.dim() == 4 and probs.shape[1] > 1 and (not channelcnt or channelcnt == probs.shape[1])
    if not probs.is_cuda or not plain:
        if probs.is_cuda:
            raise NotImplementedError("the HIP path takes [N,C,H,W] scores with C > 1 and [N,H,W] integer labels")
        return _host_dice_old(probs, act, is_list_bat, nosft, channelcnt)
    _lib.lib()
    sq = _as_square_labels(act)
    if sq.dtype != torch.int64:
        raise RuntimeError("scatter(): Expected dtype int64 for index")
    z = probs.detach().contiguous()
    if z.dtype not in (torch.float32, torch.bfloat16):
        z = z.float()
    labels = sq.contiguous()
    base = NO_SFT if nosft else 0
    s0, iflags = _dice_sums(z, labels, base)
    s1, _ = _dice_sums(z, labels, base | SELF_ADJ)
    dice = (2 * s0[..., 0] + 1) / (s0[..., 1] + s1[..., 1] + s0[..., 2] + 1)
    kept = dice[:, 1:]
    if is_list_bat:
        return kept
    sel = iflags.bool()
    nsel = sel.sum()
    div = torch.where(nsel > 0, nsel * kept.shape[1], torch.tensor(kept.numel(), device=z.device))
    return (kept * sel[:, None]).sum() / div


def argmax_mask(logits):
    """uint8 arg-max mask [N,H,W] of NCHW scores (first maximum wins, like torch.argmax)."""
    _lib.lib()
    z = logits.detach().contiguous()
    N, C = z.shape[0], z.shape[1]
    HW = z[0, 0].numel()
    mask = torch.empty((N,) + tuple(z.shape[2:]), dtype=torch.uint8, device=z.device)
    counts = torch.empty(N, C, 3, dtype=torch.float32, device=z.device)
    iflags = torch.empty(N, dtype=torch.int32, device=z.device)
    labels = torch.zeros(N, HW, dtype=torch.int64, device=z.device)
    dev, st = _env(z)
    call("segk_argmax_dice", z.data_ptr(), dt(z.dtype), labels.data_ptr(), N, C, HW, mask.data_ptr(), counts.data_ptr(),
         iflags.data_ptr(), dev, st)
    return mask


# ------------------------------------------------------------------------------------ public: focal
def focal_loss_with_logits(p, t, gamma=2., alpha=None, reduction='elementwise_mean'):
    if not p.is_cuda:
        return _host_focal(p, t, gamma, alpha, reduction)
    if t.dtype != torch.int64:
        raise RuntimeError(f"expected scalar type Long but found {t.dtype}")
    if reduction not in ('elementwise_mean', 'none', 'sum'):
        raise ValueError('Reduction parameter unknown.')
    if reduction == 'none':
        if torch.is_grad_enabled() and p.requires_grad:
            # the per-pixel map with a gradient: element-wise torch ops (differentiable like the reference's map,
            # loss/focal.py:64-75); the fused kernels serve the reduced losses of the training loops and the map without grad
            return _host_focal(p, t, gamma, alpha, reduction)
        out = dice_focal(p, t, gamma=gamma, alpha=alpha, want_map=True)
        return out[2].to(p.dtype)
    _, loss = dice_focal(p, t, gamma=gamma, alpha=alpha, focal_sum=(reduction == 'sum'))
    return loss.to(p.dtype)


def reduce(to_reduce, reduction):
    """reference loss/focal.py:5-30"""
    if reduction == 'elementwise_mean':
        return torch.mean(to_reduce)
    if reduction == 'none':
        return to_reduce
    if reduction == 'sum':
        return torch.sum(to_reduce)
    raise ValueError('Reduction parameter unknown.')


def _general_focal_loss(p, t, gamma, loss_val, alpha_weight=1., reduction='elementwise_mean'):
    """Focal weighting of an arbitrary per-pixel loss (reference loss/focal.py:32-75): p probabilities, t one-hot target.
    Element-wise torch glue on whatever device the tensors live on; the training path is focal_loss_with_logits, which
    fuses softmax, cross entropy and this weighting into the HIP loss kernels."""
    weight = (1 - torch.where(torch.eq(t, 1.), p, 1 - p)).pow(gamma)
    if loss_val.dim() < weight.dim():
        loss_val = loss_val.unsqueeze(1)
    return reduce(weight * alpha_weight * loss_val, reduction)


def _focal_loss(p, t, gamma, loss_val, reduction):
    """reference loss/focal.py:77-106: integer labels -> one-hot, class weight 1."""
    onehot = torch.zeros_like(p).scatter_(1, t.unsqueeze(1), 1)
    return _general_focal_loss(p=p, t=onehot, gamma=gamma, loss_val=loss_val, reduction=reduction, alpha_weight=1.)


# ------------------------------------------------------------------------------------ CPU tensors

# ------------------------------------------------------------------------------------ public: edge-aware Dice
class _EdgeFn(torch.autograd.Function):
    """fp32 [N,C,H,W] -> edge map |IN(sobel_x)| + |IN(sobel_y)| per channel (segk_edge_fwd / segk_edge_bwd)."""

    @staticmethod
    def forward(ctx, x):
        _lib.lib()
        xf = x.contiguous().float()
        N, C, H, W = xf.shape
        dev, st = _env(xf)
        edge = torch.empty_like(xf)
        stats = torch.empty(N * C * 4, dtype=torch.float32, device=xf.device)
        ws = workspace(query("segk_edge_workspace", N * C, H, W), xf.device)
        call("segk_edge_fwd", xf.data_ptr(), N * C, H, W, 1e-5, edge.data_ptr(), stats.data_ptr(), ws.data_ptr(), ws.numel(), dev, st)
        ctx.save_for_backward(xf, stats)
        ctx.in_dtype = x.dtype
        return edge

    @staticmethod
    def backward(ctx, dedge):
        xf, stats = ctx.saved_tensors
        N, C, H, W = xf.shape
        dev, st = _env(xf)
        g = dedge.contiguous().float()
        dx = torch.empty_like(xf)
        ws = workspace(query("segk_edge_bwd_workspace", N * C, H, W), xf.device)
        call("segk_edge_bwd", xf.data_ptr(), g.data_ptr(), stats.data_ptr(), N * C, H, W, dx.data_ptr(), ws.data_ptr(), ws.numel(), dev, st)
        return dx.to(ctx.in_dtype)


def get_hot_enc(input, channels=3):
    """One-hot planes of a label image (reference loss/edge_loss.py:32-55; 3-D input is reshaped with shape[2] twice, as there)."""
    if input.dim() == 2:
        input = input.unsqueeze(0).unsqueeze(0)
    if input.dim() == 3:
        input = input.view(input.shape[0], 1, input.shape[2], input.shape[2])
    zer = torch.zeros(input.shape[0], channels, *input.shape[2:], device=input.device)
    return zer.scatter(1, input.long(), 1)


def get_edg_c(tens):
    """Per-channel edge map (reference loss/hausdorff.py:71-108): Sobel derivatives in both directions and both signs,
    each through InstanceNorm2d and ReLU, summed."""
    if tens.dim() == 3:
        tens = tens.float().unsqueeze(1)
    if tens.is_cuda:
        return _EdgeFn.apply(tens)
    return _host_edges(tens)


def edge_conv_loss(probs, act):
    """Squared-Dice loss between the edge maps of the prediction and of the one-hot target (reference
    loss/hausdorff.py:110-128; three classes, as get_hot_enc's default there)."""
    return dice_loss_sq(get_edg_c(probs), get_edg_c(get_hot_enc(act)))


def _host_edges(x):
    """CPU tensors: the derivative pair as one grouped convolution with two filters per channel; the opposite-sign pair
    of the reference is the negation, and relu(IN(s)) + relu(IN(-s)) = |IN(s)|."""
    C = x.shape[1]
    pair = torch.tensor([[[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]], [[1., 2., 1.], [0., 0., 0.], [-1., -2., -1.]]], dtype=x.dtype)
    s = F.conv2d(x, pair.unsqueeze(1).repeat(C, 1, 1, 1).to(x.device), padding=1, groups=C)      # [N, 2C, H, W]: (x, y) per channel
    u = F.instance_norm(s, eps=1e-5).abs()
    return u[:, 0::2] + u[:, 1::2]


def _host_onehot(labels, n_classes):
    idx = _as_square_labels(labels)
    return torch.zeros(idx.shape[0], n_classes, idx.shape[2], idx.shape[3]).scatter(1, idx, 1)


def _host_fg_mean(table, act, keep_all, need_multi):
    kept = table if (keep_all or table.shape[1] == 1) else table[:, 1:]
    rows = torch.unique(torch.nonzero(act, as_tuple=True)[0])
    denom = kept.numel()
    if len(rows) and (table.shape[1] > 1 or not need_multi):
        denom = len(rows) * kept.shape[1]
    return kept[rows].sum() / denom, kept


def _host_dice_sq(scores, act, no_sft, keep_all, self_adjusting=False):
    g = act if act.dim() == 4 else _host_onehot(act, scores.shape[1])
    p = scores if no_sft else scores.softmax(1)
    if self_adjusting:
        wgt = 1 - p
        table = (2 * (p * g * wgt).sum((2, 3)) + 1) / ((wgt * p).sum((2, 3)) + g.sum((2, 3)) + 1)
    else:
        table = (2 * (p * g).sum((2, 3)) + 1) / ((p * p).sum((2, 3)) + (g * g).sum((2, 3)) + 1)
    return _host_fg_mean(table, act, keep_all, True)[0]


def _host_dice_metric(scores, act, as_list, nosft, channelcnt, keep_all):
    if act.dim() != 4:
        channelcnt = channelcnt or scores.shape[-3]
        g = _host_onehot(act, channelcnt)
    else:
        g = act
    p = scores if nosft else scores.softmax(1)
    if p.dim() != 3:
        p = p.argmax(1)
    p = p.unsqueeze(1)
    ph = torch.zeros(p.shape[0], channelcnt, *p.shape[-2:]).scatter(1, p.long(), 1)
    table = (2 * (ph * g).sum((2, 3)) + 1) / (ph.sum((2, 3)) + g.sum((2, 3)) + 1)
    mean, kept = _host_fg_mean(table, act, keep_all, False)
    return kept if as_list else mean


def _host_focal(p, t, gamma, alpha, reduction):
    ce = F.cross_entropy(p, t, weight=alpha, reduction='none').unsqueeze(1)
    q = p.softmax(1)
    is_t = torch.zeros_like(q, dtype=torch.bool).scatter_(1, t.unsqueeze(1), True)
    weight = (1 - torch.where(is_t, q, 1 - q)).pow(gamma)
    val = weight * ce
    if reduction == 'elementwise_mean':
        return val.mean()
    if reduction == 'sum':
        return val.sum()
    if reduction == 'none':
        return val
    raise ValueError('Reduction parameter unknown.')


def _host_dice_old(scores, act, as_list, nosft, channelcnt):
    act = act.view(act.shape[0], 1, act.shape[2], act.shape[2])
    g = torch.zeros(act.shape[0], channelcnt or scores.shape[1], *act.shape[2:], device=act.device).scatter(1, act, 1)
    if scores.dim() == 4 and scores.shape[-3] == 1:      # a [N,1,H,W] label map instead of scores
        channelcnt = channelcnt or int(act.max()) + 1
        p = torch.zeros(scores.shape[0], channelcnt, *scores.shape[2:], device=scores.device).scatter(1, scores.long(), 1)
    else:
        p = scores if nosft else scores.softmax(1)
    table = (2 * (p * g).sum((2, 3)) + 1) / (p.sum((2, 3)) + g.sum((2, 3)) + 1)
    kept = table[:, 1:]
    if as_list:
        return kept
    rows = torch.unique(torch.nonzero(act, as_tuple=True)[0])
    denom = len(rows) * kept.shape[1] if len(rows) else kept.numel()
    return kept[rows].sum() / denom

"""Worker of tests/test_zz_gpu_data_parallel.py: one of two ranks that share the GPU (gloo rendezvous).  Each rank runs one
training step of a small U-Net on its own batch twice -- once alone (plain autograd gradients), once through FlatModel
with the bucketed all-reduce -- and checks that the data-parallel gradients are the mean over ranks of the local ones, that
all ranks started from rank 0's weights, and that the fused Adam step leaves all ranks with identical parameters."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import UNetSimple
    from oracle.ref_models import hash_init_, synthetic_batch
    from segk.flat import FlatModel
    from segk.optim import Adam

    x, t = synthetic_batch(2, 32, 3, in_ch=1, tag=f"dp:rank{rank}")
    x, t = x.cuda(), t.cuda()

    def step(model):
        logits = model(x)
        (dice_loss_sq(logits, t) + focal_loss_with_logits(logits, t)).backward()

    alone = hash_init_(UNetSimple(channelscale=16)).cuda().train()
    step(alone)
    torch.manual_seed(7 + rank)                                  # ranks start apart: rank 0's weights must win
    model = UNetSimple(channelscale=16)
    if rank == 0:
        hash_init_(model)
    model = model.cuda().train()
    flat = FlatModel(model, bucket_mb=0.05)
    assert len(flat.buckets.ranges) >= 3
    for (k, p), (_, q) in zip(model.named_parameters(), alone.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), f"rank {rank}: {k} was not broadcast from rank 0"
    opt = Adam(flat, lr=1e-3)
    step(model)
    torch.cuda.synchronize()
    for (k, p), (_, q) in zip(model.named_parameters(), alone.named_parameters()):
        mine = q.grad.detach().cpu()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        want = sum(both) / world
        got = p.grad.detach().cpu()
        err = float((got - want).abs().max())
        assert err <= 1e-6 + 1e-5 * float(want.abs().max()), f"rank {rank}: {k}: {err}"
    opt.step()
    torch.cuda.synchronize()
    w = flat.flat_param.detach().cpu()
    both = [torch.zeros_like(w) for _ in range(world)]
    dist.all_gather(both, w)
    assert torch.equal(both[0], both[1]), "ranks diverged after the optimizer step"
    print(f"rank {rank}: data-parallel gradients and parameters agree", flush=True)
    vgg_against_the_oracle(rank, world)
    dist.destroy_process_group()


def vgg_against_the_oracle(rank, world):
    """UNetVgg itself, against the parity oracle of the data-parallel path: the CPU oracle run on every rank's own shard,
    gradients averaged over the ranks.  The shards are the two well-conditioned batches (oracle/wellcond.py), on which
    fp32 gradients are reproducible to 1e-5, so the all-reduced HIP gradients must meet 1e-4."""
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import UNetVgg
    from oracle import ref_losses as RL
    from oracle import wellcond as WC
    from oracle.ref_models import RefUNetVgg, hash_init_
    from segk.flat import FlatModel
    x, t = WC.batch(WC.TAGS[rank % len(WC.TAGS)])
    ref = WC.condition_(hash_init_(RefUNetVgg(out_classes=4))).train()
    lr = ref(x)
    (RL.dice_loss_sq(lr, t) + RL.focal_loss_with_logits(lr, t)).backward()
    net = WC.condition_(hash_init_(UNetVgg(out_classes=4))).cuda().train()
    flat = FlatModel(net)
    assert len(flat.buckets.ranges) >= 4
    lg = net(x.cuda())
    (dice_loss_sq(lg, t.cuda()) + focal_loss_with_logits(lg, t.cuda())).backward()
    torch.cuda.synchronize()
    worst = 0.0
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        mine = q.grad.detach()
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        want = (sum(both) / world).double()
        scale = float(want.norm())
        if scale < 1e-6:      # analytically zero gradients: rounding noise
            continue
        err = float((p.grad.detach().cpu().double() - want).norm()) / scale
        worst = max(worst, err)
        assert err <= 1e-4, f"rank {rank}: {k}: rel-L2 {err:.3e} against the shard-averaged oracle gradients"
    print(f"rank {rank}: UNetVgg all-reduced gradients match the shard-averaged oracle (worst rel-L2 {worst:.2e})", flush=True)


if __name__ == "__main__":
    main()

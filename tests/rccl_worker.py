"""Worker of tests/test_zz_gpu_data_parallel.py::test_rccl_branch_single_rank: ONE rank with the `nccl` (= RCCL) backend on
the one GPU, collectives forced on (SEGK_FORCE_COLLECTIVES=1), so that the RCCL branch of the bucketed all-reduce -- event
on the compute stream, wait on the communication stream, `all_reduce` on device memory, scaling, event back -- executes
for real, with the weight gradients on their side stream as in every other run (three streams: compute, weight gradients,
communication).  With one rank the reduction (an average over the ranks) is the identity: gradients and the Adam update must
equal those of a plain run bit for bit."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-segmentation_amd"))
sys.path.insert(0, ROOT)


def main():
    os.environ["SEGK_FORCE_COLLECTIVES"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetvgg import UNetSimple
    from oracle.ref_models import hash_init_, synthetic_batch
    from segk.flat import FlatModel
    x, t = synthetic_batch(2, 32, 3, in_ch=1, tag="rccl:one-rank")
    x, t = x.cuda(), t.cuda()

    def step(model):
        logits = model(x)
        (dice_loss_sq(logits, t) + focal_loss_with_logits(logits, t)).backward()

    plain = hash_init_(UNetSimple(channelscale=16)).cuda().train()
    step(plain)
    model = hash_init_(UNetSimple(channelscale=16)).cuda().train()
    flat = FlatModel(model, bucket_mb=0.05)
    assert flat.buckets.active and flat.buckets.comm_stream is not None and dist.get_backend() == "nccl"
    seen = []
    real = dist.all_reduce

    def counting(tensor, *a, **k):
        seen.append((tensor.is_cuda, torch.cuda.current_stream() == flat.buckets.comm_stream))
        return real(tensor, *a, **k)
    dist.all_reduce = counting
    for _ in range(2):
        step(model)
    torch.cuda.synchronize()
    dist.all_reduce = real
    from segk import engine
    # the weight gradients ran on the side stream: the bucket all-reduces waited for two producer streams next to their own
    assert engine._WGRAD_STREAM and engine.side_stream(torch.device("cuda", 0), create=False) is not None
    n = len(flat.buckets.ranges)
    assert n >= 3 and len(seen) == 2 * n and all(cuda and on_comm for cuda, on_comm in seen), (n, seen[:4])
    for (k, p), (_, q) in zip(model.named_parameters(), plain.named_parameters()):
        assert torch.equal(p.grad, q.grad), k            # identity reduction: bit for bit
    print(f"rank 0: RCCL branch ran {len(seen)} bucket all-reduces on the communication stream; gradients unchanged", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

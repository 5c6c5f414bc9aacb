"""World-size-2 CPU (gloo) test of the data-parallel path: flat parameter/gradient buffers and bucketed
all-reduce launched as backward completes gradients.  The tape hooks are driven by hand in the order the GPU
tape would fire them (reverse registration order); the result must equal the mean of the per-rank gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "image-segmentation_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from models.unetplusplus import conv_block_nested
    from segk.flat import FlatModel
    torch.manual_seed(100 + rank)                      # different initial weights per rank: rank 0's must win
    net = conv_block_nested(8, 16, 8)
    flat = FlatModel(net, bucket_mb=0.0001)            # ~100 B buckets -> several buckets even for this tiny block
    assert len(flat.buckets.ranges) >= 3
    w0 = torch.cat([p.detach().reshape(-1) for _, p in list(net.named_parameters())[::-1]])
    gathered = [torch.zeros_like(w0) for _ in range(world)]
    dist.all_gather(gathered, w0)
    assert torch.equal(gathered[0], gathered[1]), "initial weights were not broadcast from rank 0"
    sink, alloc, begin, finish = net._segk_grad_hooks[:4]
    names = [n for n, _ in net.named_parameters()]
    local = {}
    for step in range(2):                              # two backward passes: bucket state must reset
        begin()
        g = torch.Generator().manual_seed(1000 * step + rank)
        for n in reversed(names):                      # backward completion order
            p = dict(net.named_parameters())[n]
            buf = alloc(n, p)
            assert buf.data_ptr() == p.grad.data_ptr()  # gradients are written straight into the flat buffer
            local[n] = torch.randn(p.shape, generator=g)
            buf.copy_(local[n])
            sink(n, buf)
        finish()
        assert sorted(flat.buckets.launched) == list(range(len(flat.buckets.ranges)))
        for n in names:                                # expected: mean over ranks of what each rank wrote
            mine = local[n]
            both = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            want = sum(both) / world
            got = dict(net.named_parameters())[n].grad
            assert torch.allclose(got, want, rtol=1e-6, atol=1e-7), (n, step)
    q.put((rank, "ok"))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_allreduce(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(world)) == list(range(world))


def test_bucket_partition_covers_flat_buffer():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "image-segmentation_amd"))
    from segk.flat import GradBuckets
    sizes = [("a", 1000), ("b", 10), ("c", 5000), ("d", 7), ("e", 3)]
    b = GradBuckets(sizes, bucket_bytes=4 * 1030)
    # every tensor starts on a 64-element (256-byte) boundary: 1024 + 64 + 5056 + 64 + 64
    assert b.ranges[0][0] == 0 and b.ranges[-1][1] == 6272 == b.total
    assert all(off % 64 == 0 for off, _ in b.offsets.values()) and b.offsets["c"] == (1088, 5000)
    for (lo, hi), (lo2, _) in zip(b.ranges, b.ranges[1:]):
        assert hi == lo2
    assert b.bucket_of["a"] == b.bucket_of["b"] == 0 and b.bucket_of["c"] == 1 and b.bucket_of["e"] == 2
    with pytest.raises(AssertionError):
        b.finish()                                     # nothing was marked: incomplete buckets are an error


def _wrapper_worker(rank, world, port, q):
    """A wrapper module around two tape networks (the shape of UNetVggTwo / SegmModelPL) plus a frozen parameter."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "image-segmentation_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from models.unetplusplus import conv_block_nested
    from segk.flat import FlatModel

    class Two(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.left = conv_block_nested(8, 8, 8)
            self.right = conv_block_nested(8, 8, 8)

    torch.manual_seed(7 + rank)
    net = Two()
    net.right.bn2.bias.requires_grad_(False)            # frozen: neither bucketed nor all-reduced
    flat = FlatModel(net, bucket_mb=0.0005)
    assert "right.bn2.bias" not in flat.views and "left.conv1.weight" in flat.views
    assert net.left._segk_grad_hooks is not None and net.right._segk_grad_hooks is not None
    expect = {}
    # autograd runs the two tapes' backward passes one after the other: right first (it was used last), then left
    for sub_name in ("right", "left"):
        sub = getattr(net, sub_name)
        sink, alloc, begin, finish = sub._segk_grad_hooks[:4]
        begin()
        g = torch.Generator().manual_seed(31 * rank + len(sub_name))
        for n, p in reversed(list(sub.named_parameters())):
            buf = alloc(n, p)
            val = torch.randn(p.shape, generator=g)
            buf.copy_(val)
            if p.requires_grad:
                assert buf.data_ptr() == p.grad.data_ptr()
                expect[sub_name + "." + n] = val
            else:
                assert p.grad is None
            sink(n, buf)
        finish()                                         # only the last network's finish waits for the buckets
    assert sorted(flat.buckets.launched) == list(range(len(flat.buckets.ranges)))
    for n, val in expect.items():
        both = [torch.zeros_like(val) for _ in range(world)]
        dist.all_gather(both, val)
        assert torch.allclose(dict(net.named_parameters())[n].grad, sum(both) / world, rtol=1e-6, atol=1e-7), n
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_flat_model_over_a_wrapper_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_wrapper_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


class _TapeStandIn(torch.autograd.Function):
    """What segk.net._TapeFn does with the hooks of a data-parallel wrapper, on CPU tensors: the forward is the module's
    plain torch path, the backward produces every parameter gradient through alloc/sink between begin() and finish()."""

    @staticmethod
    def forward(ctx, net, x, *params):
        hooks = net._segk_grad_hooks
        hooks[4]()
        with torch.enable_grad():
            xin = x.detach()
            out = net._forward_host(xin)
        ctx.net, ctx.out, ctx.hooks = net, out, hooks
        return out.detach()

    @staticmethod
    def backward(ctx, dout):
        sink, alloc, begin, finish = ctx.hooks[:4]
        begin()
        named = [(n, p) for n, p in ctx.net.named_parameters()]
        grads = torch.autograd.grad(ctx.out, [p for _, p in named if p.requires_grad], dout)
        by_name = dict(zip([n for n, p in named if p.requires_grad], grads))
        for n, p in reversed(named):                       # backward completion order
            if p.requires_grad:
                buf = alloc(n, p)
                buf.copy_(by_name[n])
                sink(n, buf)
        finish()
        return (None, None) + tuple(None for _ in named)


def _partial_worker(rank, world, port, q):
    """Three tape networks under one wrapper: one fully frozen (records no backward), one used, one whose output never
    reaches the loss (its backward is never called).  The pass must still close: used gradients averaged, unused zero."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "image-segmentation_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from models.unetplusplus import conv_block_nested
    from segk.flat import FlatModel

    class Three(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.frozen = conv_block_nested(8, 8, 8)
            self.used = conv_block_nested(8, 8, 8)
            self.unused = conv_block_nested(8, 8, 8)

    torch.manual_seed(3)
    net = Three().train()
    for p in net.frozen.parameters():
        p.requires_grad_(False)
    flat = FlatModel(net, bucket_mb=0.0005)
    assert flat._ntargets == 2                               # the frozen network is not waited for
    run = lambda sub, x: _TapeStandIn.apply(sub, x, *sub.parameters())
    for step in range(3):
        g = torch.Generator().manual_seed(17 * step + rank)
        x = torch.randn(2, 8, 6, 6, generator=g)
        if step == 1:                                        # a backward pass cut short by an exception left the state open:
            flat._begin()                                    # the next forward must start a clean step
            flat._queued = True
            assert flat._open
        with torch.no_grad():
            h = net.frozen._forward_host(x)                  # need_grad is False: no tape, no backward node
        y = run(net.used, h)
        _ = run(net.unused, h)                               # forward only
        for p in net.unused.parameters():
            p.grad.fill_(123.0)                              # stale values that the pass must overwrite with zeros
        y.square().mean().backward()
        assert not flat._open and sorted(flat.buckets.launched) == list(range(len(flat.buckets.ranges))), step
        for n, p in net.unused.named_parameters():
            assert torch.count_nonzero(p.grad) == 0, (step, n)
        for n, p in net.used.named_parameters():
            mine = p.grad.clone()
            both = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(both, mine)
            assert torch.equal(both[0], both[1]), (step, n)  # every rank holds the same averaged gradient
            assert torch.count_nonzero(mine) > 0 or "conv" in n and "bias" in n, (step, n)
    q.put((rank, "ok"))
    dist.destroy_process_group()


def test_flat_model_closes_the_pass_when_networks_are_frozen_or_unused():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_partial_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


def test_flat_model_rejects_modules_without_a_tape_network():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "image-segmentation_amd"))
    from segk.flat import FlatModel
    with pytest.raises(TypeError, match="SegkNet"):
        FlatModel(torch.nn.Linear(4, 4))


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must run two ranks (rendezvous-only dry run on the CPU),
    and a rank count that contradicts --gpus must be an error, not a silently mislabelled one-rank run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["batch_per_gpu"] == 32 and line["global_batch"] == 64
    # --strong: the batch is the global one, divided over the ranks; --bucket-mb reaches the ranks
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--strong", "--bucket-mb", "10"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["batch_per_gpu"] == 16 and line["global_batch"] == 32 and line["bucket_mb"] == 10
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--strong", "--batch", "33"],
                       capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode != 0 and "does not divide" in r.stderr
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True,
                       text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr

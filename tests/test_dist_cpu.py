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
    sink, alloc, begin, finish = net._segk_grad_hooks
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


def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5)[0] for _ in range(2)) == [0, 1]


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

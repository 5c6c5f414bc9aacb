"""Throughput of the segmentation training step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): UNet-VGG (VGG16-BN encoder, bilinear decoder, 4 classes) on synthetic
384x384 slices, 32 slices per GPU, bf16 kernels with fp32 accumulation/statistics, squared-Dice + focal loss.
One step = forward + fused loss + backward (+ bucketed RCCL gradient all-reduce when N > 1) + fused Adam.
Prints ONE JSON line on rank 0: whole-job slices/s, the MFMA roofline of the convolution kernels (HIP-event
timed on the launch stream during the timed steps) and a CPU baseline of the oracle on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "image-segmentation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

ALGO_CONV_GB_PER_STEP = 22.81  # every convolution operand (bf16 activations of fwd, dgrad, wgrad) moved once, batch 32
GFLOP_PER_SLICE = 509.72       # conv fwd+dgrad+wgrad of UNetVgg(out_classes=4) at 384x384, 2 FLOP/MAC, no dgrad on conv1.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MI355X matrix peaks


def synthetic(batch, side, n_classes, device, seed=20):
    """z-scored grey slice replicated to 3 channels; label map with a few small ellipses, 1 in 8 slices empty."""
    g = torch.Generator().manual_seed(seed)
    grey = torch.randn(batch, 1, side, side, generator=g)
    x = grey.expand(batch, 3, side, side).contiguous()
    yy, xx = torch.meshgrid(torch.arange(side), torch.arange(side), indexing="ij")
    t = torch.zeros(batch, side, side, dtype=torch.int64)
    for b in range(batch):
        if b % 8 == 7:
            continue
        for c in range(1, n_classes):
            cy, cx = side * (0.4 + 0.07 * c), side * (0.35 + 0.1 * c)
            ry, rx = side * (0.04 + 0.01 * (b % 3)), side * 0.07
            t[b][((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = c
    return x.to(device), t.to(device)


def detectmid_gflop(model, depth, side):
    """Convolution GFLOP per volume of DetectMid (fwd + dgrad + wgrad, 2 FLOP/MAC, no dgrad on the first convolution)."""
    d, h, w = depth, side // 2, side // 2                       # after `first` (stride 1,2,2)
    macs = model.first.weight.numel() * d * h * w               # counted 2x below (fwd + wgrad)
    total = 2 * 2 * macs
    h, w = h // 2, w // 2                                       # after the DenseNet stem (stride 1,2,2)
    total += 3 * 2 * model.rsnet.features.conv1.weight.numel() * d * h * w
    d, h, w = (d - 1) // 2 + 1, (h - 1) // 2 + 1, (w - 1) // 2 + 1   # MaxPool3d(3, 2, 1)
    for name, mod in model.rsnet.features.named_children():
        if name.startswith("denseblock"):
            for layer in mod.children():
                total += 3 * 2 * (layer.conv1.weight.numel() + layer.conv2.weight.numel()) * d * h * w
        elif name.startswith("transition"):
            total += 3 * 2 * mod.conv.weight.numel() * d * h * w
            d, h, w = d // 2, h // 2, w // 2
    return total / 1e9


def host_cpu():
    """(threads this process may use, os.cpu_count(), CPU model string).  The GPU box hands a one-GPU job a share of
    the host's cores and says so through OMP_NUM_THREADS; os.cpu_count() there is the whole machine."""
    total = os.cpu_count() or 1
    share = total
    for var in ("OMP_NUM_THREADS", "MAX_JOBS"):
        v = os.environ.get(var, "")
        if v.isdigit() and int(v) > 0:
            share = min(share, int(v))
            break
    if hasattr(os, "sched_getaffinity"):
        share = min(share, len(os.sched_getaffinity(0)))
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return share, total, model


def cpu_baseline(n_classes, warmup=3, timed=10, seconds_budget=45.0):
    """The CPU oracle (plain torch.nn restatement pinned to the reference) on configs[0]: N=4, fp32, fwd+loss+bwd, with
    k = every core this job may use and k = 8 threads; 3 warm-up + 10 timed steps each, median (fewer timed steps only
    if the whole leg would run past `seconds_budget`, and then the sample says how many)."""
    from oracle import ref_losses as RL
    from oracle.ref_models import RefUNetVgg
    share, total, cpu_model = host_cpu()
    torch.manual_seed(20)
    model = RefUNetVgg(out_classes=n_classes).train()
    x, t = synthetic(4, 384, n_classes, "cpu")

    def one():
        t0 = time.perf_counter()
        model.zero_grad()
        logits = model(x)
        loss = RL.dice_loss_sq(logits, t) + RL.focal_loss_with_logits(logits, t)
        loss.backward()
        return time.perf_counter() - t0

    runs = {}
    ks = [share] + ([8] if share != 8 else [])
    t_start = time.perf_counter()
    for i, k in enumerate(ks):
        torch.set_num_threads(k)
        deadline = t_start + seconds_budget * (i + 1) / len(ks)
        for _ in range(warmup):
            one()
        times = []
        for _ in range(timed):
            times.append(one())
            if time.perf_counter() > deadline and len(times) >= 3:
                break
        med = sorted(times)[len(times) // 2]
        runs[k] = {"slices_per_s": round(4 / med, 4), "threads": k, "timed_steps": len(times), "warmup_steps": warmup,
                   "median_s_per_step": round(med, 4)}
    main = runs[share]
    return {"value": main["slices_per_s"], "unit": "slices/s", "cores": share, "kind": "port",
            "sample": f"oracle RefUNetVgg fp32, batch 4 x 3x384x384, fwd+loss+bwd, median of {main['timed_steps']} timed steps after "
                      f"{warmup} warm-up, torch.set_num_threads({share})",
            "os_cpu_count": total, "cpu_model": cpu_model, "threads_available_to_job": share,
            "runs": {f"k={k}": v for k, v in runs.items()}}


def cpu_baseline_detectmid(depth, side):
    """The CPU oracle of the 3-D classifier (RefDetectMid, pinned to the reference) on ONE volume, fp32, fwd+loss+bwd."""
    from oracle.ref_models import RefDetectMid
    threads = min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    torch.manual_seed(20)
    model = RefDetectMid(in_classes=1).train()
    x = torch.randn(2, 1, depth, side, side)       # BatchNorm needs more than one value per channel at the last stage
    t = torch.tensor([depth // 2, depth // 3])
    times = []
    for _ in range(2):
        t0 = time.perf_counter()
        model.zero_grad()
        torch.nn.functional.cross_entropy(model(x), t).backward()
        times.append(time.perf_counter() - t0)
    return {"value": round(2 / times[-1], 4), "unit": "volumes/s", "cores": threads, "kind": "port",
            "sample": f"oracle RefDetectMid fp32, batch 2 x 1x{depth}x{side}x{side}, fwd+loss+bwd, second of 2 steps"}


def csrc_hash():
    """sha256 over the kernel sources (the stamp scratch/hbm_traffic.py puts on a traffic measurement)."""
    import hashlib
    d = os.path.join(ROOT, "image-segmentation_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start `torch.distributed.run` with N ranks of this script as a
    CHILD process (this process has made no GPU call and makes none), relay its output -- rank 0's JSON line is the
    last line on stdout -- and return its exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None,
                    help="slices (volumes) per GPU; default: 32 for the UNet-VGG headline configuration, 16 for UNet++ and UNetSimple, "
                         "8 for SpiderNet, 12 for the 3-D classifier (the batch sizes of BASELINE.json and of the reference's scripts)")
    ap.add_argument("--side", type=int, default=384)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--model", default="unetvgg", choices=["unetvgg", "unetvgg_convtrans", "unetpp", "detectmid", "unetsimple", "spidernet"])
    ap.add_argument("--depth", type=int, default=64, help="slices per volume (detectmid)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    ap.add_argument("--deep-sup", action="store_true",
                    help="unetpp only: four supervised heads on x(0,1)..x(0,4), loss = sum over heads of Dice + focal (BASELINE configs[3])")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --batch is the GLOBAL batch, split over the ranks")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test: rendezvous of the N ranks on the CPU (gloo), no GPU work, prints the rank count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-wgrad-stream", action="store_true",
                    help="keep the weight gradients on the main stream (default: on a second HIP stream next to the data gradients, "
                         "for every rank count; the steps whose launches are event-timed keep everything on one stream)")
    ap.add_argument("--bucket-mb", type=float, default=25.0,
                    help="size of the flat gradient buckets that are all-reduced one by one while the backward pass is still running")
    ap.add_argument("--launch-table", default=None,
                    help="also write the event-timed duration of every convolution launch of a step (mean over the event-timed steps, "
                         "in launch order, with its shape) to this JSON file")
    args = ap.parse_args()
    if os.environ.get("SEGK_BENCH_TRACE_AFTER"):      # debugging aid: dump every thread's Python stack and exit if still running after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["SEGK_BENCH_TRACE_AFTER"]), exit=True)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # before anything touches the GPU: the ranks are child processes
    # Rank 0 prints exactly one line on stdout.  Libraries write there too (RCCL prints a version banner when its first
    # communicator is made), so file descriptor 1 is pointed at stderr for the run and the line goes to a copy of the real one.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                 f"(or run `python bench.py --gpus {args.gpus}` directly, which starts the ranks itself)")
    if args.batch is None:
        # UNet-VGG 32 and UNet++ 16 as BASELINE.json quotes them; the 3-D classifier at the reference's own 12 (train_midlslice.py:24)
        args.batch = {"unetpp": 16, "unetsimple": 16, "spidernet": 8, "detectmid": 12}.get(args.model, 32)
    if args.strong:
        if args.batch % world:
            sys.exit(f"bench.py --strong: global batch {args.batch} does not divide over {world} ranks")
        args.batch //= world
    if args.dry_run:
        import torch.distributed as dist
        seen = 1
        if world > 1:
            dist.init_process_group("gloo")
            one = torch.ones(1)
            dist.all_reduce(one)
            seen = int(one.item())
            dist.destroy_process_group()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"dry_run": True, "n_gpus": seen, "steps": args.steps, "warmup": args.warmup, "scaling": "strong" if args.strong else "weak",
                              "batch_per_gpu": args.batch, "global_batch": seen * args.batch, "bucket_mb": args.bucket_mb}), file=real_stdout, flush=True)
        return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)     # rehearsals may put several ranks on one card (gloo only)
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    forced = world == 1 and os.environ.get("SEGK_FORCE_COLLECTIVES") == "1"
    if forced:
        # a one-rank RCCL group whose bucket all-reduces are really issued (identity): what the collective machinery costs
        # per step by itself -- events, the communication stream, one RCCL launch per bucket -- without any inter-GPU transfer
        import socket
        import torch.distributed as dist
        if "MASTER_PORT" not in os.environ:
            sk = socket.socket()
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            sk.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=0, world_size=1, **({"device_id": device} if args.backend == "nccl" else {}))

    from loss.dice import dice_loss_sq
    from loss.focal import focal_loss_with_logits
    from models.unetplusplus import deep_supervision_loss
    from models.unetvgg import UNetVgg
    from segk import ops
    from segk.flat import FlatModel
    from segk.optim import Adam

    n_classes = 4
    torch.manual_seed(20)
    if args.model == "detectmid":
        # BASELINE.json configs[4]: 3-D DenseNet-169 middle-slice classifier on 64 x 256 x 256 volumes
        from models.resnt import DetectMid
        model = DetectMid(in_classes=1)
        if args.side == 384:
            args.side = 256
        gflop = detectmid_gflop(model, args.depth, args.side)
    elif args.model == "unetpp":
        from models.unetplusplus import Nested_UNet
        model = Nested_UNet(in_ch=1, out_ch=n_classes, deep_supervision=args.deep_sup)
        gflop = 1859.61                            # 3x3 convolutions; the 1x1 heads run as streaming kernels
    elif args.model == "unetsimple":
        # the configuration the reference trains (roi_train/train_roi.py:39): one input channel, channelscale 128
        from models.unetvgg import UNetSimple
        model = UNetSimple(in_classes=1, out_classes=n_classes, channelscale=128)
        gflop = None                               # taken from the FLOP count the convolution launches report
    elif args.model == "spidernet":
        # the configuration of perlosstrain/train_percloss.py:39 (channelscale 128), logits only
        from models.unetvgg import SpiderNet
        model = SpiderNet(in_classes=1, out_classes=n_classes, channelscale=128)
        gflop = None
    else:
        model = UNetVgg(out_classes=n_classes, if_convtrans=args.model == "unetvgg_convtrans")
        gflop = GFLOP_PER_SLICE if args.model == "unetvgg" else 539.05
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    model = model.to(device).train().set_compute_dtype(dtype)
    flat = FlatModel(model, bucket_mb=args.bucket_mb)      # flat params/grads; bucketed all-reduce when world > 1
    opt = Adam(flat, lr=1e-4)
    if args.model == "detectmid":
        g = torch.Generator().manual_seed(20 + rank)
        x = torch.randn(args.batch, 1, args.depth, args.side, args.side, generator=g).to(device)
        t = torch.randint(0, min(args.depth, 80), (args.batch,), generator=g).to(device)
    else:
        x, t = synthetic(args.batch, args.side, n_classes, device, seed=20 + rank)
    if args.model in ("unetpp", "unetsimple", "spidernet"):
        x = x[:, :1].contiguous()

    def step():
        logits = model(x)
        if args.model == "detectmid":
            loss = torch.nn.functional.cross_entropy(logits, t)      # train_midlslice.py:29
        elif isinstance(logits, list):                                # deep supervision: summed over the heads
            loss = deep_supervision_loss(logits, t, [dice_loss_sq, focal_loss_with_logits])
        else:
            loss = dice_loss_sq(logits, t) + focal_loss_with_logits(logits, t)
        loss.backward()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    from segk import engine
    engine._WGRAD_STREAM = engine._WGRAD_STREAM and not args.no_wgrad_stream
    overlap = engine._WGRAD_STREAM
    for _ in range(args.warmup):
        step()
    barrier()
    # Per-launch HIP events (the roofline leg) put a marker on the queue around every convolution launch, which costs
    # ~2 % of the step; they are recorded on every 4th timed step only, inside the timed region.
    timer = None if args.no_roofline else []
    timed_steps = 0
    if flat.buckets.active:
        flat.buckets.exposed = []              # events around the compute stream's wait for the collectives, every step
    t0 = time.perf_counter()
    for i in range(args.steps):
        sample = timer is not None and i % 4 == 0
        ops.CONV_TIMER = timer if sample else None
        ops.STREAM_TIMING = i % 8 == 0                    # the HBM-bound passes on every second event-timed step only
        engine._WGRAD_STREAM = overlap and not sample     # two convolutions sharing the GPU would stretch each other's event times
        timed_steps += int(sample)
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ops.CONV_TIMER = None
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt)
    final_loss = float(loss.detach())
    exposed_ms = None
    if flat.buckets.exposed:
        exposed_ms = sum(a.elapsed_time(b) for a, b in flat.buckets.exposed) / len(flat.buckets.exposed)
    flat.buckets.exposed = None
    overhead_ms = None
    if forced:                                   # the same steps with the collectives switched off: the difference is their fixed cost
        k3 = max(5, min(10, args.steps))
        times = {}
        for active in (True, False, True, False):
            flat.buckets.active = active
            step()
            barrier()
            t2 = time.perf_counter()
            for _ in range(k3):
                step()
            barrier()
            times.setdefault(active, []).append((time.perf_counter() - t2) / k3 * 1e3)
        flat.buckets.active = True
        overhead_ms = min(times[True]) - min(times[False])
    # second number: forward + loss + backward (+ all-reduce) only, no optimizer step
    k2 = max(3, min(10, args.steps))
    engine._WGRAD_STREAM = overlap
    barrier()
    t1 = time.perf_counter()
    for _ in range(k2):
        lg = model(x)
        if args.model == "detectmid":
            l2 = torch.nn.functional.cross_entropy(lg, t)
        elif isinstance(lg, list):
            l2 = deep_supervision_loss(lg, t, [dice_loss_sq, focal_loss_with_logits])
        else:
            l2 = dice_loss_sq(lg, t) + focal_loss_with_logits(lg, t)
        l2.backward()
    barrier()
    no_opt = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([no_opt], device=device, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        no_opt = float(tt)
    # host -> device copy of one batch from pinned memory (informational: never part of `value`, inputs are resident)
    hx, ht = x.detach().cpu().pin_memory(), t.cpu().pin_memory()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        x.copy_(hx, non_blocking=True)
        t.copy_(ht, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    h2d_ms = e0.elapsed_time(e1) / 5

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = world * args.batch * args.steps / elapsed
        out = {
            "metric": ("train volumes/sec (fwd+bwd+loss) DenseNet3D-169 DetectMid " + f"{args.depth}x{args.side}x{args.side}") if args.model == "detectmid" else
                      "train slices/sec (fwd+bwd+loss) " + {"unetvgg": "UNet-VGG", "unetvgg_convtrans": "UNet-VGG (transposed-conv decoder)",
                                                              "unetpp": "UNet++ deep-supervision (4 heads)" if args.deep_sup else "UNet++", "unetsimple": "UNetSimple (channelscale 128)",
                                                              "spidernet": "SpiderNet (channelscale 128)"}[args.model] + f" {args.side}x{args.side}",
            "value": round(value, 2), "unit": "volumes/s" if args.model == "detectmid" else "slices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": (f"detectmid {args.depth}x{args.side}x{args.side}, 80 classes, {args.batch} volumes/GPU, fwd + cross entropy + bwd + "
                                    if args.model == "detectmid" else
                                    f"{args.model}{' deep supervision: 4 heads on x(0,1)..x(0,4), summed Dice/focal' if args.deep_sup else ''} {args.side}x{args.side}, "
                                    f"{n_classes} classes, {args.batch} slices/GPU, fwd + fused Dice/focal loss + bwd + ")
                                   + f"{('bucketed ' + ('RCCL' if args.backend == 'nccl' else args.backend) + ' all-reduce + ') if world > 1 else ''}fused Adam",
                       "global_batch": world * args.batch, "parallelism": f"dp{world}", "random_init_weights": True,
                       "final_loss": round(final_loss, 5), "h2d_ms_per_batch": round(h2d_ms, 3),
                       "slices_per_s_without_optimizer_step": round(world * args.batch * k2 / no_opt, 2),
                       "collective": ({"backend": torch.distributed.get_backend(), "ranks": torch.distributed.get_world_size(),
                                       "op": "all-reduce (average) of the flat fp32 gradient, one call per bucket on a communication stream, "
                                             "launched when the backward pass has written the bucket's last gradient",
                                       "bucket_mb_requested": args.bucket_mb,
                                       "bucket_bytes": [4 * (hi - lo) for lo, hi in flat.buckets.ranges],
                                       "exposed_wait_ms_per_step": None if exposed_ms is None else round(exposed_ms, 3),
                                       "weight_gradients_on_side_stream": bool(engine._WGRAD_STREAM and (world == 1 or engine._WGRAD_STREAM_DP)),
                                       "one_rank_forced": forced,
                                       "one_rank_overhead_ms_per_step": None if overhead_ms is None else round(overhead_ms, 3)}
                                      if flat.buckets.active else None),
                       "value_if_h2d_not_overlapped": round(world * args.batch * args.steps / (elapsed + args.steps * h2d_ms / 1e3), 2)},
        }
        if timer:
            # What a pair of event markers costs by itself: the elapsed time of pairs with NOTHING between them, on the same
            # stream.  Every convolution launch of the event-timed steps is bracketed by such a pair, so its reading is the
            # kernel's duration plus this much marker latency; the roofline uses the durations with it taken off and reports
            # the raw sum next to them (the rocprofv3 kernel durations under profiles/ are the cross-check).
            pairs = []
            for _ in range(200):
                m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                m0.record()
                m1.record()
                pairs.append((m0, m1))
            torch.cuda.synchronize()
            marker_ms = sorted(m0.elapsed_time(m1) for m0, m1 in pairs)[len(pairs) // 2]
            streams = [(variant, max(a.elapsed_time(b) - marker_ms, 0.0), nbytes) for k, a, b, nbytes, variant, _ in timer if k == "stream"]
            timer = [r for r in timer if r[0] != "stream"]
            raw = [(k, a.elapsed_time(b), fl, variant, desc) for k, a, b, fl, variant, desc in timer]
            dur = [(k, max(ms_ - marker_ms, 0.0), fl, variant, desc) for k, ms_, fl, variant, desc in raw]
            raw_ms = sum(ms_ for _, ms_, *_ in raw) / timed_steps
            fwd = sum(ms_ for k, ms_, *_ in dur if k == "conv_fwd")
            wg = sum(ms_ for k, ms_, *_ in dur if k == "conv_wgrad")
            # the same events per kernel family, with the FLOP each launch actually multiplied (the 27-value patch row of the
            # first convolution counts as its 32 stored channels, so the sum is 0.1 % above the algorithmic figure)
            fam = {}
            for k, t_, fl, variant, _ in dur:
                name = "conv_wgrad_kernel" if k == "conv_wgrad" else f"conv_fwd_kernel[{variant}]"
                ms_, fl_, n_ = fam.get(name, (0.0, 0.0, 0))
                fam[name] = (ms_ + t_, fl_ + fl, n_ + 1)
            conv_ms = (fwd + wg) / timed_steps
            if args.launch_table:
                per = len(timer) // timed_steps
                rows = []
                for i in range(per):
                    k, _, _, fl, variant, desc = timer[i]
                    us = [1e3 * dur[s_ * per + i][1] for s_ in range(timed_steps)]
                    rows.append({"i": i, "kind": k, "variant": variant, "shape": desc, "gflop": round(fl / 1e9, 2),
                                 "us_mean": round(sum(us) / len(us), 1), "us_min": round(min(us), 1), "us_max": round(max(us), 1)})
                with open(args.launch_table, "w") as f:
                    json.dump(rows, f, indent=0)
            if gflop is None:      # FLOP as launched (channel padding included), per slice
                gflop = sum(fl for _, _, _, fl, _, _ in timer) / timed_steps / args.batch / 1e9
            achieved = gflop * args.batch / conv_ms        # GFLOP / ms = TFLOP/s
            peak = MFMA_PEAK_TFLOPS[args.dtype]
            # HBM bytes of the same launches (one step) from the committed PMC passes of this workload (counters cannot be read
            # from inside this process): reported only while the kernel sources are the ones the passes were taken on
            traffic, traffic_note = None, "no PMC passes are committed for this workload"
            tfile = os.path.join(ROOT, "profiles", "r03_hbm_traffic.json")
            if args.model == "unetvgg" and args.dtype == "bf16" and args.batch == 32 and args.side == 384 and os.path.exists(tfile):  # noqa: E501
                with open(tfile) as f:
                    tdoc = json.load(f)
                if tdoc.get("csrc_sha256") == csrc_hash():
                    traffic = tdoc.get("conv_mfma_kernels_hbm_bytes_per_step")
                    traffic_note = (f"HBM bytes per step of these launches (rocprofv3 PMC, profiles/r03_hbm_traffic.json, taken on these kernel "
                                    f"sources); algorithmic minimum {ALGO_CONV_GB_PER_STEP} GB")
                else:
                    traffic_note = ("profiles/r03_hbm_traffic.json was measured on other kernel sources (csrc hash differs): not reported; "
                                    "re-run scratch/profile_round.sh")
            out["roofline"] = {"bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                               "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
                               "kernel": "conv_fwd_kernel + conv_wgrad_kernel (all MFMA convolution launches of a step)",
                               "launches_per_step": len(timer) // timed_steps, "conv_ms_per_step": round(conv_ms, 3),
                               "conv_ms_per_step_raw_events": round(raw_ms, 3), "event_marker_pair_us": round(1e3 * marker_ms, 2),
                               "event_timed_steps": timed_steps,
                               "fwd_dgrad_ms": round(fwd / timed_steps, 3), "wgrad_ms": round(wg / timed_steps, 3),
                               "whole_step_mfma_frac": round(gflop * args.batch / ms / peak, 4),
                               "by_kernel": {name: {"launches_per_step": n_ // timed_steps, "ms_per_step": round(ms_ / timed_steps, 3),
                                                    "achieved": round(fl_ / ms_ / 1e9, 1), "frac": round(fl_ / ms_ / 1e9 / peak, 4)}
                                             for name, (ms_, fl_, n_) in sorted(fam.items()) if ms_ > 0}}
            # the HBM-bound passes of the same event-timed steps: algorithmic bytes / event-timed duration, by family
            HBM_PEAK_GBS = 8000.0
            sfam = {}
            for name, ms_, nbytes in streams:
                t_, b_, n_ = sfam.get(name, (0.0, 0.0, 0))
                sfam[name] = (t_ + ms_, b_ + nbytes, n_ + 1)
            ssteps = len(range(0, args.steps, 8))
            out["roofline"]["streams"] = {name: {"calls_per_step": n_ // ssteps, "ms_per_step": round(t_ / ssteps, 3),
                                                 "algorithmic_gb_per_step": round(b_ / ssteps / 1e9, 3),
                                                 "achieved_gbs": round(b_ / t_ / 1e6, 1), "frac_of_hbm_peak": round(b_ / t_ / 1e6 / HBM_PEAK_GBS, 4)}
                                          for name, (t_, b_, n_) in sorted(sfam.items()) if t_ > 0}
            out["roofline"]["streams_note"] = ("bound hbm, peak 8000 GB/s; a call is one pass family launch group (e.g. the two passes and the "
                                               "finalize of a BatchNorm backward), HIP-event timed on the launch stream with the marker latency taken off")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_detectmid(args.depth, args.side) if args.model == "detectmid" else cpu_baseline(n_classes)
            if args.model not in ("unetvgg", "detectmid"):
                out["cpu_baseline"]["sample"] += " (the headline UNet-VGG workload, not this model)"
        print(json.dumps(out), file=real_stdout, flush=True)
    if world > 1 or forced:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

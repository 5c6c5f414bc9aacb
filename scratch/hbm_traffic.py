"""HBM bytes per kernel launch from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one program run each):
    python scratch/hbm_traffic.py fetch_counter_collection.csv write_counter_collection.csv STEPS > profiles/...json
Both counters are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes, so reads are doubled."""
import collections
import csv
import hashlib
import json
import os
import re
import sys


def csrc_hash():
    """sha256 over the kernel sources the measurement belongs to (bench.py reports the traffic only while it matches)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-segmentation_amd", "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()


def per_kernel(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024)
    return acc


fetch, write = per_kernel(sys.argv[1]), per_kernel(sys.argv[2])
# STEPS = auto: the fused loss forward runs exactly once per step
steps = (float(sys.argv[3]) if sys.argv[3] != "auto" else
         float(sum(len(v) for k, v in fetch.items() if "dice_focal_fwd_kernel" in k)))
kernels, conv, total = {}, 0.0, 0.0
for name in sorted(fetch, key=lambda k: -sum(fetch[k]) * 2 - sum(write.get(k, []))):
    rd, wr = 2 * sum(fetch[name]), sum(write.get(name, []))
    n = len(fetch[name])
    total += rd + wr
    # what bench.py times as convolution; the statistics-only and sums passes of the fused first layer stand in for BatchNorm passes
    norm_pass = re.search(r"conv_first_fwd_kernel<\d+, \d+, 1>|conv_first_wgrad_kernel<\d+, 1>", name) is not None
    if any(k in name for k in ("conv_fwd_kernel", "conv_direct_kernel", "conv_wgrad_kernel", "conv_wgrad_reduce", "conv_first_")) and not norm_pass:
        conv += rd + wr
    kernels[name] = {"launches_per_step": round(n / steps, 2), "hbm_read_bytes_per_launch": int(rd / n),
                     "hbm_write_bytes_per_launch": int(wr / n)}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 2 --warmup 1`; "
                     "KiB counters; FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B)",
           "steps_profiled": steps,
           "csrc_sha256": csrc_hash(),
           "conv_mfma_kernels_hbm_bytes_per_step": int(conv / steps),
           "all_kernels_hbm_bytes_per_step": int(total / steps),
           "kernels": kernels}, sys.stdout, indent=1)

"""Per-launch table of one training step: every kernel of a steady-state step from a rocprofv3 kernel trace (start offset,
duration, gap to the previous kernel's end), and for the convolution launches the in-step duration next to the sustained
(back-to-back, same kernel for tens of milliseconds) duration of the same layer.

    python scratch/instep_table.py TRACE_kernel_trace.csv LAUNCH_TABLE.json SUSTAINED.txt > profiles/r03_instep_vs_sustained.txt

TRACE: `rocprofv3 --kernel-trace --output-format csv` of `bench.py --no-wgrad-stream` (one stream, so nothing overlaps);
LAUNCH_TABLE: `bench.py --launch-table` (HIP-event durations, launch order, shapes); SUSTAINED: output of
`SUSTAIN=40 python scratch/conv_bench.py`.
"""
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("segk::", "")
    m = re.match(r"_ZN4segk\d+([A-Za-z_0-9]+?)I(.*?)EEv", name)
    if m:
        args = re.findall(r"Li(\d+)E|Lb([01])E|(DF16b|f)", m.group(2))
        flat = [a or b or c for a, b, c in args]
        return m.group(1) + "<" + ",".join(flat) + ">"
    return name[:70]


def is_conv(name):
    if "conv_wgrad_reduce" in name or "conv_first_wgrad_reduce" in name or "conv_pack" in name:
        return None
    if ("conv_fwd_kernel" in name or "conv_direct_kernel" in name) and "conv_first" not in name:
        return "conv_fwd"
    if "conv_wgrad_kernel" in name and "conv_first" not in name:
        return "conv_wgrad"
    s = short(name)
    if "conv_first_fwd_kernel" in name and re.search(r"[, ]2>$", s):
        return "conv_fwd"
    if "conv_first_wgrad_kernel" in name and re.search(r"[, ]2>$", s):
        return "conv_wgrad"
    return None


def main():
    trace, table, sustained = sys.argv[1:4]
    rows = []
    with open(trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # steps end with the fused Adam launch; take the steady steps (skip warm-up: use the last 6 complete ones)
    ends = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    steps = [rows[ends[k] + 1: ends[k + 1] + 1] for k in range(len(ends) - 1)]
    n = max(set(len(s) for s in steps), key=[len(s) for s in steps].count)
    steps = [s for s in steps if len(s) == n][-6:]
    print(f"# {len(steps)} steady steps of {n} kernels each (one stream); times in microseconds, mean over those steps")
    sus = {}
    for line in open(sustained):
        m = re.match(r"(\S+)\s+(\d+)\^2\s+(\d+)->\s*(\d+)\s+[\d.]+ GF \|(.*)", line)
        if m:
            H, ci, co = int(m.group(2)), int(m.group(3)), int(m.group(4))
            for which, us in re.findall(r"(fwd|dgrad|wgrad)\s+([\d.]+) us", m.group(5)):
                sus[(H, ci, co, which)] = (float(us), m.group(1))
    launches = json.load(open(table))
    conv_i = 0
    tot_in = tot_ev = tot_sus = tot_gap = 0.0
    nfwd = sum(1 for l in launches if l["kind"] == "conv_fwd")
    n_forward = (nfwd + 1) // 2            # forward convolutions lead the step; every one but the first has a data gradient
    seen_fwd = 0
    print(f"{'#':>3} {'kernel':58s} {'start':>9} {'dur':>8} {'gap':>6} | {'layer':9s} {'shape':22s} {'event':>8} {'sustained':>9} {'in/sus':>6}")
    t0s = [s[0][0] for s in steps]
    step_len = sum(s[-1][1] - s[0][0] for s in steps) / len(steps) / 1e3
    for i in range(n):
        dur = sum(s[i][1] - s[i][0] for s in steps) / len(steps) / 1e3
        start = sum(s[i][0] - t0 for s, t0 in zip(steps, t0s)) / len(steps) / 1e3
        gap = sum((s[i][0] - s[i - 1][1]) for s in steps) / len(steps) / 1e3 if i else 0.0
        tot_gap += gap
        name = steps[0][i][2]
        kind = is_conv(name)
        extra = ""
        if kind and conv_i < len(launches):
            l = launches[conv_i]
            conv_i += 1
            m = re.match(r"(\d+)\^2 (\d+)->(\d+) k(\d+)", l["shape"] or "")
            s_us, lname = None, ""
            if m and l["kind"] == kind:
                H, a, b = int(m.group(1)), int(m.group(2)), int(m.group(3))
                if kind == "conv_wgrad":
                    key = (H, a, b, "wgrad")
                elif seen_fwd < n_forward:
                    key = (H, a, b, "fwd")
                else:
                    key = (H, b, a, "dgrad")
                if key in sus:
                    s_us, lname = sus[key]
            if kind == "conv_fwd":
                seen_fwd += 1
            tot_in += dur
            tot_ev += l["us_mean"]
            if s_us:
                tot_sus += s_us
            extra = f"| {lname:9s} {l['kind'][5:] + ' ' + (l['shape'] or l['variant']):22s} {l['us_mean']:8.1f} " + (f"{s_us:9.1f} {dur / s_us:6.2f}" if s_us else f"{'':9s} {'':6s}")
        print(f"{i:3d} {short(name):58s} {start:9.1f} {dur:8.1f} {gap:6.1f} {extra}")
    print(f"# step (first kernel start -> Adam end) {step_len:.1f} us; sum of gaps between consecutive kernels {tot_gap:.1f} us")
    print(f"# convolution launches: in-step (trace) {tot_in:.1f} us, HIP events {tot_ev:.1f} us, sustained (layers with a figure) {tot_sus:.1f} us; "
          f"matched {conv_i} of {len(launches)} event-timed launches")


if __name__ == "__main__":
    main()

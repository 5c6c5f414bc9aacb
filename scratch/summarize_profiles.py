"""profiles/<tag>_summary.md from one profiling round (scratch/profile_round.sh):
    python scratch/summarize_profiles.py gpurun_out/prof_round r02
Per kernel: launches per step, average duration (kernel trace), HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE) ->
achieved GB/s, and for the MFMA kernels the share of cycles in which the matrix pipes were busy
(SQ_VALU_MFMA_BUSY_CYCLES summed over the 1024 SIMDs / (1024 x GRBM_GUI_ACTIVE per XCD))."""
import collections
import csv
import json
import re
import sys

d, tag = sys.argv[1], sys.argv[2]
# steps in the trace: the fused loss forward runs exactly once per step
steps_trace = sum(int(r["Calls"]) for r in csv.DictReader(open(f"{d}/trace_kernel_stats.csv")) if "dice_focal_fwd_kernel" in r["Name"])


def pmc(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch, write, mfma = pmc(f"{d}/fetch_counter_collection.csv"), pmc(f"{d}/write_counter_collection.csv"), pmc(f"{d}/mfma_counter_collection.csv")
rows = list(csv.DictReader(open(f"{d}/trace_kernel_stats.csv")))
bench = json.loads(open(f"{d}/bench_n1.json").read().strip().splitlines()[-1])
out = [f"# Profile summary {tag}: `bench.py` (N=1), rocprofv3 kernel trace + PMC passes", "",
       f"bench line: {bench['value']} {bench['unit']}, {bench['ms_per_step']} ms/step; roofline: {json.dumps(bench.get('roofline'))}", "",
       "| kernel | launches/step | avg µs | ms/step | HBM MB/launch (read+write) | GB/s | % of 8 TB/s | MFMA busy |", "|---|---|---|---|---|---|---|---|"]
total = 0.0
for r in rows:
    name = r["Name"]
    ms = float(r["TotalDurationNs"]) / 1e6 / steps_trace
    total += ms
    if ms < 0.02:
        continue
    avg_us = float(r["AverageNs"]) / 1e3
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void |segk::", "", short)[:70]
    f = fetch.get(name, {}).get("FETCH_SIZE")
    w = write.get(name, {}).get("WRITE_SIZE")
    mb = gbps = pct = ""
    if f and w:
        per = (2 * sum(f) / len(f) + sum(w) / len(w)) * 1024
        mb = f"{per / 1e6:.1f}"
        gbps = f"{per / (avg_us * 1e-6) / 1e9:.0f}"
        pct = f"{100 * per / (avg_us * 1e-6) / 8e12:.0f}"
    busy = ""
    m = mfma.get(name)
    if m and sum(m.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) > 0:
        b, g = sum(m["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(m["GRBM_GUI_ACTIVE"])
        busy = f"{100 * b / (1024 * g / 8):.0f} %"
    out.append(f"| `{short}` | {int(r['Calls']) / steps_trace:.1f} | {avg_us:.1f} | {ms:.3f} | {mb} | {gbps} | {pct} | {busy} |")
out += ["", f"sum of kernel time: {total:.2f} ms/step"]
open(f"profiles/{tag}_summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))

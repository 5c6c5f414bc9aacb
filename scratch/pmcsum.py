"""Average PMC counter values per kernel from rocprofv3 counter_collection CSVs."""
import csv, sys, collections, re
for f in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"])[:70]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "conv" not in k or "pack" in k: continue
        print(k, "n=", len(next(iter(d.values()))))
        for c, v in d.items():
            v = v[2:] if len(v) > 2 else v
            print(f"    {c:28s} {sum(v)/len(v):16.0f}")

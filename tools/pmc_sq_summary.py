"""Per-kernel means of the rocprofv3 counter_collection CSVs given on the command line (values in millions)."""
import csv, sys, collections, re
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    per_dispatch = collections.defaultdict(float)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:28]
            per_dispatch[(k, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, _, c), v in per_dispatch.items(): acc[k][c].append(v)
    for k in sorted(acc):
        def fmt(c, v):
            m = sum(v) / len(v)
            return f"{c.replace('SQ_', '')}={m / 1e6:,.1f}M" if m >= 1e6 else f"{c.replace('SQ_', '')}={m:,.0f}"
        print(f"{k:28s} " + " ".join(fmt(c, v) for c, v in sorted(acc[k].items())))

"""Summarise rocprofv3 --pmc counter_collection CSVs: median counter value per kernel."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = r['Kernel_Name'].replace("void ", "")[:44]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, cs in sorted(agg.items()):
            if k.startswith("k_fill") or k.startswith("__amd"): continue
            print(f"{k:44s} " + "  ".join(f"{c}={sorted(v)[len(v)//2]:.4g}" for c, v in sorted(cs.items())))

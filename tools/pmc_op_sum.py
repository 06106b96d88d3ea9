"""sum a rocprofv3 --pmc counter over all compute kernels of a run (k_fill / copy kernels excluded), per op.
usage: python tools/pmc_op_sum.py <dir> <counter> <ops>"""
import collections, csv, glob, sys
d, counter, ops = sys.argv[1], sys.argv[2], float(sys.argv[3])
tot = collections.defaultdict(float)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace("void ", "")
        if r['Counter_Name'] != counter or k.startswith("k_fill") or k.startswith("__amd"):
            continue
        tot[k.split("(")[0][:40]] += float(r['Counter_Value'])
s = sum(tot.values())
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"{k:42s} {v / ops:12.0f} " + ("KiB per op" if counter.endswith("SIZE") else "per op"))
print(f"{counter} total {s / ops:.0f} KiB per op = {s / ops * 1024 / 1e6:.1f} MB" if counter.endswith("SIZE") else f"{counter} total {s / ops:.0f} per op")

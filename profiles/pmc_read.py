import csv, glob, sys, collections
root = sys.argv[1]; pat = sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            agg[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} avg={sum(v)/len(v):.4g}")

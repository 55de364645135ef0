import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2])
agg = collections.defaultdict(list)
tot = 0
for r in rows:
    name = r['Kernel_Name']
    d = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot += d
    key = (name[:70], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg[key].append(d)
print(f"total kernel time per step: {tot/nsteps/1e3:.2f} ms")
byname = collections.defaultdict(float)
for k, v in agg.items(): byname[k[0]] += sum(v)
print("--- by kernel name (ms/step) ---")
for k, v in sorted(byname.items(), key=lambda kv: -kv[1])[:22]:
    print(f"{v/nsteps/1e3:8.3f}  {k}")
print("--- top launches ---")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[3]) if len(sys.argv)>3 else 25]:
    print(f"{sum(v)/nsteps:9.1f} us/step  n/step={len(v)/nsteps:4.1f} avg={sum(v)/len(v):8.1f}us  {k}")

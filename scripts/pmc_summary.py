"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files."""
import csv, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"_ZN3hvc12_GLOBAL__N_1\d+", "", r["Kernel_Name"])[:40]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print("==", k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} avg={sum(v)/len(v):16.1f}")

"""dev helper: per-kernel mean of every counter in rocprofv3 counter_collection.csv files under a directory."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"], r["Kernel_Name"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, k, c), v in per.items():
        acc[k][c].append(v)
pat = sys.argv[2:] or [""]
for k in acc:
    if not any(p in k for p in pat): continue
    print(k[:70])
    for c, v in sorted(acc[k].items()):
        print("   %-28s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))

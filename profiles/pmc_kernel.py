"""Sum the PMC counters of the kernels whose name contains argv[2] from a rocprofv3 --pmc output dir argv[1]
(per-launch averages)."""
import csv, glob, sys
from collections import defaultdict
acc, n = defaultdict(float), defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:32s} launches {n[k]:4d}  avg {acc[k]/max(n[k],1):16.1f}")

"""print avg duration (us) of kernels whose name contains argv[2] from a rocprofv3 --stats output dir argv[1]"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us")

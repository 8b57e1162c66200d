"""The kernel sequence of the last step of a rocprofv3 --kernel-trace (argv[1] = output dir; argv[2] = fraction of the trace to
keep, default 0.5 = the last of two steps): start offset, duration, idle gap in front (all in microseconds), queue, name.
Usage: python3 profiles/sequence.py <rocprof dir> [fraction] > sequence.txt"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1.0 - frac)):]
t0 = rows[0][0]
last_end = t0
for s, e, n, q in rows:
    print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} {max(0, s - last_end) / 1e3:8.1f} q{q:>3s} {n.split('(')[0][:70]}")
    last_end = max(last_end, e)

"""Where the stream sits idle: from a rocprofv3 --kernel-trace CSV (argv[1] = output dir), the busy time, the idle time and
the idle time attributed to the kernel that FOLLOWS each gap, for the last `steps` (argv[2], default 1) repetitions of the
workload.  A step is delimited by the first kernel of landmarks() (argv[3], default "gather_rows_kernel" is not unique, so
the script simply takes the last fraction 1/(steps+warmup) of the trace by kernel count when argv[3] is absent).

Usage: python3 profiles/timeline_gaps.py <rocprof dir> [fraction_of_trace_to_keep=0.5]
"""
import csv
import glob
import sys
from collections import defaultdict

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1.0 - frac)):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gap_after = defaultdict(lambda: [0, 0])
dur = defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = max(0, s1 - e0)
    key = n1.split("(")[0][:60]
    gap_after[key][0] += g
    gap_after[key][1] += 1
for s, e, n in rows:
    key = n.split("(")[0][:60]
    dur[key][0] += e - s
    dur[key][1] += 1
print(f"kernels {len(rows)}  span {span/1e6:.3f} ms  busy {busy/1e6:.3f} ms  idle {(span-busy)/1e6:.3f} ms")
print(f"{'kernel':60s} {'launches':>8s} {'busy ms':>9s} {'idle before, ms':>16s} {'avg gap us':>11s}")
for k, (g, n) in sorted(gap_after.items(), key=lambda kv: -kv[1][0])[:30]:
    print(f"{k:60s} {n:8d} {dur[k][0]/1e6:9.3f} {g/1e6:16.3f} {g/n/1e3:11.1f}")

"""Per kernel, from a rocprofv3 --pmc output directory (argv[1]) holding SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES: the fractions of the wave cycles spent parked at s_waitcnt /
barriers (wait_any), stalled at issue (wait_inst) and issuing (active), and MFMA-busy per busy cycle."""
import csv, glob, sys
from collections import defaultdict
acc, n = defaultdict(lambda: defaultdict(float)), defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[k] += 1
print(f"{'kernel':50s} {'n':>4s} {'wavecyc':>10s} wait_any wait_inst active mfma/busy")
for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"])[:22]:
    a = acc[k]
    w = max(a["SQ_WAVE_CYCLES"], 1.0)
    print(f"{k:50s} {n[k]:4d} {w:10.3g} {a['SQ_WAIT_ANY']/w:8.2f} {a['SQ_WAIT_INST_ANY']/w:9.2f} {a['SQ_ACTIVE_INST_ANY']/w:6.2f} "
          f"{a['SQ_VALU_MFMA_BUSY_CYCLES']/max(a['SQ_BUSY_CYCLES'],1.0):9.2f}")

set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/r4e; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-back-to-back > $OUT/trace.log 2>&1; echo "rc=$?"
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/r4e/trace/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "row_hash_kernel" in r["Kernel_Name"]]
seq=rows[idx[-1]:]
from collections import defaultdict
agg=defaultdict(list)
for r in seq:
    nm=r["Kernel_Name"]
    nm="rocprim_"+("merge" if "merge" in nm else "onesweep" if "onesweep" in nm else "segmented" if "segmented" in nm else "other") if "rocprim" in nm else nm.split("(")[0].replace("void ","")[:44]
    agg[nm].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
tot=0
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:34]:
    print(f"{k:46s} n={len(v):3d} total={sum(v):8.1f} us  each={[round(x) for x in v[:8]]}")
span=(int(seq[-1]["End_Timestamp"])-int(seq[0]["Start_Timestamp"]))/1e3
print("span us", span, "busy", sum(sum(v) for v in agg.values()))
PY

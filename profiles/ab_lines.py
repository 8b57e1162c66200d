import json,sys
for f in sys.argv[1:]:
    j=json.loads(open(f).read().strip().splitlines()[-1])
    k=j.get("kernels",{})
    print(f, round(j["ms_per_step"],3), "landmarks", round(j["phases_ms"]["landmarks"],3), "sweep", round(j["phases_ms"]["sweep"],3), {a:round(b["avg_launch_ms"],4) for a,b in k.items() if "rss2" in a or "eig" in a})

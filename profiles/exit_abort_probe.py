"""Diagnostic for the exit-time SIGSEGV of round 1 (gpurun_out/prof_r01f/stats.err): run a small score, dump
/proc/self/maps next to the profiler output and leave the process WITHOUT closing the context
(CGE_NO_ATEXIT_CLOSE=1 disables api.py's atexit hook).  Under `rocprofv3 --kernel-trace --stats -- python3 <this>` a
crash prints unsymbolised frames; the maps file says which library owns each of them.
usage: python3 profiles/exit_abort_probe.py <maps-output-file> [bench-size]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cge.jl_amd as cg  # noqa: E402
from cge.jl_amd import api, synth  # noqa: E402

big = len(sys.argv) > 2 and sys.argv[2] == "big"
g = synth.abcd_like(200000 if big else 20000, 2000000 if big else 200000, 40, 64, seed=3)
ctx = api.Context(0)
ctx.set_inputs(g["edges"], g["eweights"], g["vweights"], g["comm"], g["embedding"])
res = ctx.score(g["clusters"], 400, 4, "rss", seed=1, auc_samples=2000)
print("RES", list(res), flush=True)
with open("/proc/self/maps") as f, open(sys.argv[1], "w") as o:
    o.write(f.read())
# no ctx.close(): the context outlives the interpreter when CGE_NO_ATEXIT_CLOSE=1

"""Generates tests/golden/oracle_generated.json from the CPU ORACLE (not from the Julia reference:
Julia is not available in the build image).  Every entry is labelled with what produced it."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cge.jl_amd as cg  # noqa: E402
from conftest import random_samples  # noqa: E402
from oracle import oracle as orc  # noqa: E402

g = os.path.join(ROOT, "tests", "golden", "example10k")
a = cg.parseargs(["-g", f"{g}/10k.edgelist", "-c", f"{g}/10k.ecg", "-e", f"{g}/10k.embedding", "-l", "200", "--seed", "42"])
edges, ew, vw, comm, clusters, emb, _, land, forced, method = a[:10]
dii, lemb, lcomm, ledges, lw, lweight, v2l = orc.landmarks(edges, ew, vw, clusters, comm, emb, False, land, forced, method, False)
smp = random_samples(np.random.default_rng(42), len(ew), len(vw), 10000)
res, tr = orc.wGCL(ledges, lw, lcomm, lemb, dii, lweight, vw, v2l, edges, ew, emb, False, smp, trace=True)
out = {"_provenance": "oracle/cge_oracle.c (CPU restatement); README.md:99 pins result[0:4] of example10k_l200_rss",
       "example10k_l200_rss": {"result": list(res), "iters": tr["iters"], "div": tr["div"], "N": int(len(dii)),
                                "hi": orc.max_pair_dist(emb), "v_to_l_head": v2l[:32].tolist()}}
with open(os.path.join(ROOT, "tests", "golden", "oracle_generated.json"), "w") as f:
    json.dump(out, f, indent=1)
print(out["example10k_l200_rss"]["result"])

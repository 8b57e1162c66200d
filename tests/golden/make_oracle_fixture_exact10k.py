"""Generates tests/golden/oracle_exact10k.json: config 1 of BASELINE.json (example/10k.*, --force-exact --seed 42)
through the CPU ORACLE in true exact mode (N = n = 10 000; ~50 M packed distances, minutes of CPU).  The
sampled pairs are simple seeded draws (committed with the fixture) so that the GPU test can replay them."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import cge.jl_amd as cg  # noqa: E402
from oracle import oracle as orc  # noqa: E402

g = os.path.join(ROOT, "tests", "golden", "example10k")
a = cg.parseargs(["-g", f"{g}/10k.edgelist", "-c", f"{g}/10k.ecg", "-e", f"{g}/10k.embedding", "--force-exact",
                  "--seed", "42"])
edges, ew, vw, comm, clusters, emb, _, land = a[:8]
assert land == -1
n, S = len(vw), 10000
rng = np.random.default_rng(42)
pos = rng.integers(1, len(ew) + 1, size=(1, S))
eset = set(map(tuple, np.sort(edges, axis=1).tolist()))
ni = np.zeros((1, S), dtype=np.int64)
nj = np.zeros((1, S), dtype=np.int64)
k = 0
while k < S:
    i, j = sorted(rng.integers(1, n + 1, size=2).tolist())
    if i != j and (i, j) not in eset:
        ni[0, k], nj[0, k] = i, j
        k += 1
t0 = time.time()
res, tr = orc.wGCL(edges, ew, comm, emb, np.zeros(n), vw, [], [], np.zeros((0, 2), np.int64), [], np.zeros((0, 0)),
                   False, (pos, ni, nj), trace=True)
out = {"_provenance": "oracle/cge_oracle.c (CPU restatement), exact mode, example10k --force-exact; %.0f s" % (time.time() - t0),
       "result": list(res), "iters": tr["iters"], "div": tr["div"], "auc": tr["auc"],
       "pos_idx": pos[0].tolist(), "neg_i": ni[0].tolist(), "neg_j": nj[0].tolist()}
with open(os.path.join(ROOT, "tests", "golden", "oracle_exact10k.json"), "w") as f:
    json.dump(out, f)
print(out["result"], out["iters"], out["_provenance"])

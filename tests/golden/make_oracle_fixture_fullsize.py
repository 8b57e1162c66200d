"""Generates tests/golden/oracle_<workload>.npz: BASELINE.json's headline / config 3 / config 4 at FULL size
(10^6 vertices, 10^7 or 2*10^7 edges, d = 128, 4000 landmarks) through the CPU ORACLE (oracle/cge_oracle.c; not the
Julia reference -- there is no `julia` in the image): landmarks() with the configured split rule, then wGCL() /
wGCL_directed() in landmark mode with seeded sample draws.

The one thing the oracle cannot do at this size is the reference's O(n^2 d) point-set diameter (src/divergence.jl:104-113,
~10^14 flop): it is handed in through oracle.set_known_diameter(), computed by tests/diameter_ref.py (an exact branch and
bound in numpy whose winner is re-evaluated with dist()'s own arithmetic; checked against the oracle's loop at small
sizes in tests/test_oracle_golden.py).

The graphs are bench.py's workloads (`cge.jl_amd.synth.abcd_like(n, 1.05 m, C, d, seed=42)`), the sampled pairs
`conftest.random_samples(default_rng(42), m, n, S)`; the GPU test regenerates both, so only expected OUTPUTS are stored
(checksums for the big arrays).

usage: python tests/golden/make_oracle_fixture_fullsize.py headline|cfg3|cfg4|cfg3_size|small|cfg5_200k|cfg2_d128
"""
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cge.jl_amd import synth  # noqa: E402
from conftest import random_samples  # noqa: E402
from diameter_ref import exact_diameter  # noqa: E402
from oracle import oracle as orc  # noqa: E402

WORKLOADS = {  # == bench.py WORKLOADS
    "headline": dict(n=1_000_000, m=10_000_000, C=500, d=128, land=4000, forced=4, method="rss", samples=10000),
    "cfg3": dict(n=1_000_000, m=20_000_000, C=500, d=128, land=4000, forced=4, method="diameter", samples=10000),
    "cfg4": dict(n=1_000_000, m=10_000_000, C=500, d=128, land=4000, forced=4, method="rss", samples=1_000_000,
                 directed=True),
    "small": dict(n=50_000, m=500_000, C=25, d=128, land=200, forced=4, method="rss", samples=10000),
    # config 3's graph under the `size` rule (median cuts, src/landmarks.jl:218-238): the one rule the other full-size
    # fixtures do not reach; its diameter is config 3's (same embedding), taken from oracle_cfg3.npz when that exists
    "cfg3_size": dict(n=1_000_000, m=20_000_000, C=500, d=128, land=4000, forced=4, method="size", samples=10000,
                      same_graph_as="cfg3"),
    # config 5's own shape at the size the GPU test budget allows (tests/test_gpu_configs.py::
    # test_config5_d512_twelve_thousand_landmarks): 12 000 landmarks at d = 512, eigenvectors from LAPACK's syevr
    "cfg5_200k": dict(n=200_000, m_exact=4_200_000, C=1500, d=512, land=12000, forced=4, method="rss", samples=10000,
                      lapack=True),
    # config 2's graph with a 128-wide embedding under rss2: groups of >= 5000 rows through the d > 64 chain kernel
    "cfg2_d128": dict(n=100_000, m=1_000_000, C=50, d=128, land=400, forced=4, method="rss2", samples=10000),
}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def main(name):
    c = WORKLOADS[name]
    directed = bool(c.get("directed", False))
    t0 = time.time()
    if c.get("lapack"):
        orc.use_lapack_eig(True)  # the routine Julia's eigvecs calls; Jacobi at d = 512 is minutes per matrix
    g = synth.abcd_like(c["n"], c.get("m_exact") or int(c["m"] * 1.05), c["C"], c["d"], seed=42, directed=directed)
    print(f"[{name}] graph n={g['n']} m={g['m']} ({time.time() - t0:.0f} s)", flush=True)
    t0 = time.time()
    twin = os.path.join(ROOT, "tests", "golden", f"oracle_{c.get('same_graph_as', '')}.npz")
    if os.path.exists(twin) and int(np.load(twin)["emb_crc"]) == crc(g["embedding"]):
        fx = np.load(twin)
        hi, (hi_i, hi_j), st = float(fx["hi"]), fx["hi_pair"], "from " + os.path.basename(twin)
    else:
        hi, hi_i, hi_j, st = exact_diameter(g["embedding"], g["comm"][:, 0])
    t_hi = time.time() - t0
    print(f"[{name}] diameter {hi!r} pair ({hi_i},{hi_j}) {st} ({t_hi:.0f} s)", flush=True)
    orc.set_known_diameter(hi)
    t0 = time.time()
    dii, lemb, lcomm, ledges, lw, lweight, v2l = orc.landmarks(g["edges"], g["eweights"], g["vweights"], g["clusters"],
                                                               g["comm"], g["embedding"], False, c["land"], c["forced"],
                                                               c["method"], directed)
    t_lm = time.time() - t0
    print(f"[{name}] landmarks N={len(dii)} n_ledges={len(lw)} ({t_lm:.0f} s)", flush=True)
    smp = random_samples(np.random.default_rng(42), g["m"], g["n"], c["samples"])
    t0 = time.time()
    fn = orc.wGCL_directed if directed else orc.wGCL
    res, tr = fn(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, g["edges"], g["eweights"], g["embedding"],
                 False, smp, trace=True)
    t_sc = time.time() - t0
    print(f"[{name}] score {list(map(float, res))} iters={tr['iters']} ({t_sc:.0f} s)", flush=True)
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", f"oracle_{name}.npz"),
        provenance=np.array(f"oracle/cge_oracle.c (CPU restatement) via tests/golden/make_oracle_fixture_fullsize.py {name}; "
                            f"diameter from tests/diameter_ref.py ({t_hi:.0f} s), landmarks {t_lm:.0f} s, score {t_sc:.0f} s, "
                            "one core" + ("; eigenvectors: LAPACK syevr (oracle.use_lapack_eig), the routine Julia's eigvecs calls"
                                          if c.get("lapack") else "")),
        n=g["n"], m=g["m"], N=len(dii), v_to_l_crc=crc(v2l.astype(np.int32)), landmark_sizes=np.bincount(v2l)[1:],
        v_to_l_sample=v2l[::997].astype(np.int32), dii=dii, lweight=lweight, lcomm=lcomm[:, 0], lemb_crc=crc(lemb),
        ledges_crc=crc(ledges), lw_crc=crc(lw), n_ledges=len(lw), edges_crc=crc(g["edges"]), emb_crc=crc(g["embedding"]),
        result=res, iters=np.array(tr["iters"]), div=np.array(tr["div"]), auc=np.array(tr["auc"]), hi=hi,
        hi_pair=np.array([hi_i, hi_j]))


if __name__ == "__main__":
    main(sys.argv[1])

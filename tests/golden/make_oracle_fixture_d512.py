"""Generates tests/golden/oracle_d512.npz: the oracle pin of config 5's code paths (BASELINE.json configs[4]: d = 512,
fp32-MFMA distance) at a size the CPU ORACLE (oracle/cge_oracle.c; not the Julia reference -- there is no `julia` in the
image) finishes offline: cyclic Jacobi on a 512 x 512 covariance per split (src/landmarks.jl:160-162) bounds it.

Graph: `cge.jl_amd.synth.abcd_like(20000, 210000, 30, 512, seed=42)`; landmarks(-l 300 -f 4 -m rss), then wGCL() in
landmark mode with `conftest.random_samples(default_rng(42), m, n, 10000)`; the diameter is the oracle's own O(n^2 d)
loop (src/divergence.jl:104-113).  A second run with `-m diameter` pins the cut rules through the same wide eigen-solver.
The GPU test regenerates graph and draws, so only expected OUTPUTS are stored (v_to_l in full: 20 000 ids).

On the device this exercises what no other fixture reaches: group_eig_panel_kernel (128 < d <= 512), the tile-pair
covariance (four 128-column tiles), the K = 512 fp32-MFMA bound pass of the diameter.

usage: python tests/golden/make_oracle_fixture_d512.py [small|quick]
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
from oracle import oracle as orc  # noqa: E402

D512 = dict(n=20_000, m=200_000, C=30, d=512, land=300, forced=4, samples=10000, seed=42)
# `small` (argv[1]): the same at 8000 vertices / 12 communities / -l 120 (a third of the Jacobi problems: ~21 s each at d = 512)
D512_SMALL = dict(n=8_000, m=80_000, C=12, d=512, land=120, forced=4, samples=10000, seed=42)
# `quick` (argv[1]): every group that is split keeps >= ~500 rows.  The oracle's cyclic Jacobi takes ~26 s on the covariance of
# >= 500 rows at d = 512 and ~136 s on a rank-deficient one (150 rows), so the two fixtures above cost 4 h (small) and ~20 h
# (large) of one core per split rule; this one, 30 splits per rule, costs ~15 minutes per rule.
D512_QUICK = dict(n=12_000, m=120_000, C=6, d=512, land=36, forced=4, samples=10000, seed=42)


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else ""
    small = which == "small"
    c = D512_SMALL if small else D512_QUICK if which == "quick" else D512
    g = synth.abcd_like(c["n"], int(c["m"] * 1.05), c["C"], c["d"], seed=c["seed"])
    print(f"graph n={g['n']} m={g['m']}", flush=True)
    out = dict(n=g["n"], m=g["m"], edges_crc=crc(g["edges"]), emb_crc=crc(g["embedding"]), gen_n=c["n"], gen_m=int(c["m"] * 1.05),
               gen_C=c["C"], land=c["land"], forced=c["forced"])
    t0 = time.time()
    hi = orc.max_pair_dist(g["embedding"])
    t_hi = time.time() - t0
    print(f"hi = {hi!r} ({t_hi:.0f} s)", flush=True)
    out["hi"] = hi
    smp = random_samples(np.random.default_rng(42), g["m"], g["n"], c["samples"])
    times = {}
    for method in ("rss", "diameter"):
        t0 = time.time()
        dii, lemb, lcomm, ledges, lw, lweight, v2l = orc.landmarks(g["edges"], g["eweights"], g["vweights"],
                                                                   g["clusters"], g["comm"], g["embedding"], False,
                                                                   c["land"], c["forced"], method, False)
        t_lm = time.time() - t0
        print(f"[{method}] landmarks: N={len(dii)} ({t_lm:.0f} s)", flush=True)
        t0 = time.time()
        orc.set_known_diameter(hi)
        res, tr = orc.wGCL(ledges, lw, lcomm, lemb, dii, lweight, g["vweights"], v2l, g["edges"], g["eweights"],
                           g["embedding"], False, smp, trace=True)
        t_sc = time.time() - t0
        print(f"[{method}] wGCL: {list(map(float, res))} iters={tr['iters']} ({t_sc:.0f} s)", flush=True)
        times[method] = (t_lm, t_sc)
        p = method + "_"
        out.update({p + "N": len(dii), p + "v_to_l": v2l.astype(np.int32), p + "dii": dii, p + "lweight": lweight,
                    p + "lcomm": lcomm[:, 0], p + "lemb_crc": crc(lemb), p + "ledges_crc": crc(ledges),
                    p + "lw_crc": crc(lw), p + "n_ledges": len(lw), p + "result": res, p + "iters": np.array(tr["iters"]),
                    p + "div": np.array(tr["div"]), p + "auc": np.array(tr["auc"])})
        save(out, times, t_hi, small, which)  # after every rule: a rule of the larger variants is hours of one core


def save(out, times, t_hi, small, which):
    """The fixture with the rules finished so far (the GPU test skips a rule whose keys are missing)."""
    prov = ("oracle/cge_oracle.c (CPU restatement), tests/golden/make_oracle_fixture_d512.py; diameter "
            f"{t_hi:.0f} s; " + "; ".join(f"{m}: landmarks {t[0]:.0f} s + wGCL {t[1]:.0f} s" for m, t in times.items()) + "; one core")
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", "oracle_d512_small.npz" if small else "oracle_d512_quick.npz" if which == "quick" else "oracle_d512.npz"),
        provenance=np.array(prov), **out)


if __name__ == "__main__":
    main()

"""Generates tests/golden/oracle_d512.npz: the oracle pin of config 5's code paths (BASELINE.json configs[4]: d = 512,
fp32-MFMA distance) at a size the CPU ORACLE (oracle/cge_oracle.c; not the Julia reference -- there is no `julia` in the
image) finishes offline: cyclic Jacobi on a 512 x 512 covariance per split (src/landmarks.jl:160-162) bounds it.

Graph: `cge.jl_amd.synth.abcd_like(20000, 210000, 30, 512, seed=42)`; landmarks(-l 300 -f 4 -m rss), then wGCL() in
landmark mode with `conftest.random_samples(default_rng(42), m, n, 10000)`; the diameter is the oracle's own O(n^2 d)
loop (src/divergence.jl:104-113).  A second run with `-m diameter` pins the cut rules through the same wide eigen-solver.
The GPU test regenerates graph and draws, so only expected OUTPUTS are stored (v_to_l in full: 20 000 ids).

On the device this exercises what no other fixture reaches: group_eig_panel_kernel (128 < d <= 512), the tile-pair
covariance (four 128-column tiles), the K = 512 fp32-MFMA bound pass of the diameter.

Eigenvectors: `--eig=lapack` (the default since round 4) takes `eigvecs(A)[:, end]` from LAPACK's syevr through
scipy.linalg.eigh(driver="evr") -- the routine Julia's `eigvecs` calls (src/landmarks.jl:99,162,225,254) -- with the oracle's
sign rule; `--eig=jacobi` keeps the oracle's cyclic Jacobi (26-136 s per 512-wide matrix: only `quick` is affordable, and
oracle_d512_quick.npz was made that way).  tests/test_oracle_golden.py::test_lapack_eig_reproduces_the_jacobi_fixtures holds
the two routines to the same partitions and bits on the committed Jacobi fixtures.

`lowrank`: 6000 vertices in 20 communities (~300 rows each), -l 200 -f 4: EVERY covariance is rank-deficient (< 512 rows)
and ~120 splits happen in the global phase -- what config 5's last splits of small communities produce.

usage: python tests/golden/make_oracle_fixture_d512.py [large|small|quick|lowrank] [--eig=lapack|jacobi]
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
D512_LOWRANK = dict(n=6_000, m=60_000, C=20, d=512, land=200, forced=4, samples=10000, seed=42)
VARIANTS = {"large": (D512, "oracle_d512.npz"), "": (D512, "oracle_d512.npz"), "small": (D512_SMALL, "oracle_d512_small.npz"),
            "quick": (D512_QUICK, "oracle_d512_quick.npz"), "lowrank": (D512_LOWRANK, "oracle_d512_lowrank.npz")}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    eig = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--eig=")), "lapack")
    which = pos[0] if pos else ""
    c, fname = VARIANTS[which]
    if eig == "lapack":
        orc.use_lapack_eig(True)
    g = synth.abcd_like(c["n"], int(c["m"] * 1.05), c["C"], c["d"], seed=c["seed"])
    print(f"graph n={g['n']} m={g['m']}", flush=True)
    out = dict(n=g["n"], m=g["m"], edges_crc=crc(g["edges"]), emb_crc=crc(g["embedding"]), gen_n=c["n"], gen_m=int(c["m"] * 1.05),
               gen_C=c["C"], land=c["land"], forced=c["forced"], eig=np.array(eig))
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
        save(out, times, t_hi, fname, eig)  # after every rule: a rule of the larger variants is hours of one core


def save(out, times, t_hi, fname, eig):
    """The fixture with the rules finished so far (the GPU test skips a rule whose keys are missing)."""
    prov = (f"oracle/cge_oracle.c (CPU restatement; eigenvectors: {eig}), tests/golden/make_oracle_fixture_d512.py; diameter "
            f"{t_hi:.0f} s; " + "; ".join(f"{m}: landmarks {t[0]:.0f} s + wGCL {t[1]:.0f} s" for m, t in times.items()) + "; one core")
    np.savez_compressed(
        os.path.join(ROOT, "tests", "golden", fname),
        provenance=np.array(prov), **out)


if __name__ == "__main__":
    main()

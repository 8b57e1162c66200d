"""Static checks of the gfx950 device assembly (no GPU needed: hipcc cross-compiles).

A workgroup barrier that publishes LDS data is only safe when the publisher's stores have landed.  The compiler normally
puts `s_waitcnt lgkmcnt(0)` in front of `s_barrier`; in round 4 it left the wait out on one loop back edge of the batched
eigen-solver and the landmark pipeline started to differ from run to run.  profiles/scan_barrier_waits.py follows every path
of every kernel and reports barriers that an LDS store can reach unwaited-for; this test keeps that list empty."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cge.jl_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-munsafe-fp-atomics",
         "--cuda-device-only", "-S"]  # the Makefile's flags


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_barrier_with_an_lds_store_in_flight(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        import scan_barrier_waits as sbw
    finally:
        sys.path.pop(0)
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    assert sources

    def compile_one(src):
        out = os.path.join(tmp_path, src[:-4] + ".s")
        r = subprocess.run([HIPCC, *FLAGS, os.path.join(CSRC, src), "-o", out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return out

    with ThreadPoolExecutor(max_workers=min(8, len(sources))) as ex:
        outs = list(ex.map(compile_one, sources))
    hits = []
    n_barriers = 0
    for path in outs:
        n_barriers += open(path).read().count("s_barrier")
        hits += [(os.path.basename(path), kern, line) for kern, line in sbw.scan(path) if "rocprim" not in kern]
    shutil.rmtree(tmp_path, ignore_errors=True)
    assert n_barriers > 100  # the scan saw the kernels
    assert not hits, f"barriers reachable with an LDS store in flight: {hits}"


def test_the_scan_flags_a_missing_wait(tmp_path):
    """The checker itself: a store, a back edge to a block that starts with the barrier, no wait in between."""
    sys.path.insert(0, os.path.join(ROOT, "profiles"))
    try:
        import scan_barrier_waits as sbw
    finally:
        sys.path.pop(0)
    bad = os.path.join(tmp_path, "bad.s")
    open(bad, "w").write("""
_Z3badv:
.LBB0_1:
\ts_barrier
\tds_read_b64 v[2:3], v1
\ts_waitcnt lgkmcnt(0)
\ts_cbranch_execz .LBB0_3
\tds_write_b64 v1, v[2:3]
.LBB0_3:
\ts_cbranch_scc1 .LBB0_1
\ts_endpgm
""")
    good = os.path.join(tmp_path, "good.s")
    open(good, "w").write(open(bad).read().replace(".LBB0_3:\n", ".LBB0_3:\n\ts_waitcnt lgkmcnt(0)\n"))
    assert [k for k, _ in sbw.scan(bad)] == ["_Z3badv"]
    assert sbw.scan(good) == []

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


# kernels in which the waves of ONE workgroup hand data to each other through global memory, separated by s_barrier (the
# panel eigen-solver keeps its 2 MB matrix and the partial vectors of its tile sweep there)
GLOBAL_SHARING = ("group_eig_panel_kernel",)


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
    hits, vhits = [], []
    n_barriers = 0
    seen_shared = set()
    for path in outs:
        text = open(path).read()
        n_barriers += text.count("s_barrier")
        hits += [(os.path.basename(path), kern, line) for kern, line in sbw.scan(path) if "rocprim" not in kern]
        # kernels whose workgroup shares GLOBAL memory across its barriers: a store must have left (vmcnt(0)) before the barrier
        vhits += [(os.path.basename(path), kern, line) for kern, line in sbw.scan(path, kind="vmem", only=GLOBAL_SHARING)]
        seen_shared |= {k for k in GLOBAL_SHARING if k in text}
    shutil.rmtree(tmp_path, ignore_errors=True)
    assert n_barriers > 100  # the scan saw the kernels
    assert not hits, f"barriers reachable with an LDS store in flight: {hits}"
    assert seen_shared == set(GLOBAL_SHARING), "a kernel of the list was renamed: update GLOBAL_SHARING"
    assert not vhits, f"barriers reachable with a global store in flight, in kernels that share global scratch inside a workgroup: {vhits}"


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
    # the global-store mode: the same shape with a global store and vmcnt(0); kernels outside the list are not looked at
    vbad = os.path.join(tmp_path, "vbad.s")
    open(vbad, "w").write(open(bad).read().replace("ds_write_b64 v1, v[2:3]", "global_store_dwordx2 v[4:5], v[2:3], off"))
    vgood = os.path.join(tmp_path, "vgood.s")
    open(vgood, "w").write(open(vbad).read().replace(".LBB0_3:\n", ".LBB0_3:\n\ts_waitcnt vmcnt(0)\n"))
    assert sbw.scan(vbad) == [] and [k for k, _ in sbw.scan(vbad, kind="vmem")] == ["_Z3badv"]
    assert sbw.scan(vbad, kind="vmem", only=("other_kernel",)) == [] and sbw.scan(vgood, kind="vmem") == []

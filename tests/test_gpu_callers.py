"""Callers of the C-ABI that are not pytest-through-ctypes: the command-line script (cge_cli.py, the mirror of
example/CGE_CLI.jl:1-25), a plain C program (examples/cge_driver.c, gcc, only include/cge_hip.h) and what happens to a
context that is still alive when the interpreter exits (the exit-time abort recorded in round 1)."""
import os
import re
import shutil
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

README_VECTOR = [6.25, 0.002961243353776198, 0.0, 0.0, 9.75, 0.0017000000000000348, 0.000807441501038938]  # README.md:88-100
EX = os.path.join(GOLDEN, "example10k")
EX_FLAGS = ["-g", f"{EX}/10k.edgelist", "-c", f"{EX}/10k.ecg", "-e", f"{EX}/10k.embedding", "-l", "200", "--seed", "42"]
DRIVER = os.path.join(ROOT, "cge.jl_amd", "csrc", "build", "cge_driver")


def _vector(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry the result vector only (CGE_CLI.jl:25): {stdout!r}"
    assert re.fullmatch(r"\[[-+0-9.eInfNa, ]+\]", lines[0]), lines[0]
    return [float(x) for x in lines[0].strip("[]").replace("Inf", "inf").replace("NaN", "nan").split(",")]


def _check_readme(vec):
    # elements 1-4 are deterministic given the partition; 5-7 depend on the Julia RNG stream
    assert vec[0] == README_VECTOR[0] and vec[2] == 0.0 and vec[3] == 0.0
    assert vec[1] == pytest.approx(README_VECTOR[1], abs=1e-6)  # north star: 1e-6
    assert 0.25 <= vec[4] <= 10.0 and abs(vec[5] - README_VECTOR[5]) < 5 * README_VECTOR[6]
    assert vec[6] == pytest.approx(1.96 * np.sqrt(vec[5] * (1 - vec[5]) / 10000), rel=1e-9)


def test_julia_float_formatting():
    sys.path.insert(0, ROOT)
    import cge_cli

    assert cge_cli.julia_vector(README_VECTOR) == ("[6.25, 0.002961243353776198, 0.0, 0.0, 9.75, 0.0017000000000000348, "
                                                  "0.000807441501038938]")  # README.md:99 verbatim
    assert cge_cli.julia_float(1e-5) == "1.0e-5" and cge_cli.julia_float(1.5e22) == "1.5e22"
    assert cge_cli.julia_float(float("inf")) == "Inf" and cge_cli.julia_float(-1.0) == "-1.0"


@pytest.mark.gpu
def test_cli_script_readme_example():
    """`python cge_cli.py -g 10k.edgelist -c 10k.ecg -e 10k.embedding -l 200 --seed 42` (README.md:88-100)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "cge_cli.py")] + EX_FLAGS, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    vec = _vector(r.stdout)
    assert len(vec) == 7
    _check_readme(vec)
    dots = [l for l in r.stderr.splitlines() if l and set(l) == {"."}]
    assert dots and 1 <= len(dots[-1]) <= 40  # one "." per alpha, then a newline (src/divergence.jl:140,255)


@pytest.mark.gpu
def test_c_driver_same_call_order_same_answer():
    """The C program follows CGE_CLI.jl's call order through the ABI (set_* / landmarks_run / landmarks_fetch /
    wgcl with the init_* copies); same seed, same library-drawn samples => the same line as the Python script."""
    assert os.path.exists(DRIVER), "build the driver: make -C cge.jl_amd/csrc"
    r = subprocess.run([DRIVER] + EX_FLAGS, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    vec = _vector(r.stdout)
    _check_readme(vec)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "cge_cli.py")] + EX_FLAGS, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0 and _vector(p.stdout) == vec
    # exact mode on the reference's 115-vertex fixture, directed, weighted list
    g = os.path.join(GOLDEN, "test115")
    r = subprocess.run([DRIVER, "-g", f"{g}/test_weights.edgelist", "-c", f"{g}/test2col.ecg", "-e",
                        f"{g}/test_unordered.embedding", "-d", "--seed", "7", "--samples-local", "500"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "cge_cli.py"), "-g", f"{g}/test_weights.edgelist", "-c",
                        f"{g}/test2col.ecg", "-e", f"{g}/test_unordered.embedding", "-d", "--seed", "7",
                        "--samples-local", "500"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    # weights of 1.42 (not dyadic): the per-edge scatter adds a bin's weights in no fixed order -> last-bit differences
    vr, vp = _vector(r.stdout), _vector(p.stdout)
    assert vr[0] == vp[0] and vr[4] == vp[4] and np.allclose(vr, vp, rtol=1e-12, atol=0)


_EXIT_SCRIPT = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np
import cge.jl_amd as cg
from cge.jl_amd import api
g = os.path.join(%r, "test115")
a = cg.parseargs(["-g", g + "/test.edgelist", "-c", g + "/test1col.ecg", "-e", g + "/test_n2v.embedding", "-l", "20", "-f", "1"])
ctx = api.Context(0)
ctx.set_inputs(a[0], a[1], a[2], a[3], a[5])
res = ctx.score(a[4], 20, 1, "rss", seed=1, auc_samples=500)
print("RES", res[0])
# no ctx.close(): the context is alive when the interpreter starts to exit
"""


@pytest.mark.gpu
@pytest.mark.parametrize("profiler", [False, True])
def test_context_alive_at_interpreter_exit(tmp_path, profiler):
    """A Context that nobody closed must not take the process down at exit -- plain, and under rocprofv3 (where round 1
    recorded a SIGSEGV inside exit handlers after the tool's finalisation).  api.py closes live contexts from an `atexit`
    hook, i.e. before the C-level exit handlers of the HIP runtime / the profiler run."""
    script = tmp_path / "exit_probe.py"
    script.write_text(_EXIT_SCRIPT % (ROOT, GOLDEN))
    cmd = [sys.executable, str(script)]
    if profiler:
        rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
        if not os.path.exists(rp):
            pytest.skip("rocprofv3 not installed")
        cmd = [rp, "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(tmp_path / "prof"), "--", "python3",
               str(script)]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd="/tmp")
    assert "RES" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
    assert r.returncode == 0, f"exit code {r.returncode}\n{r.stderr[-3000:]}"
    assert "SIGSEGV" not in r.stderr and "Aborted at" not in r.stderr


@pytest.mark.gpu
def test_in_library_rccl_plumbing_one_rank():
    """csrc/collectives.cpp: librccl is found and bound at run time, a communicator is created from an id, and the three
    reductions the path uses (sum / max of doubles, sum of the words as int64) go through ncclAllReduce on the ctx stream.
    One rank only (the test box has one GPU and RCCL refuses two ranks on one device): the multi-rank semantics are those of
    ncclAllReduce, and the sharding logic around it is covered by test_gpu_two_ranks.py / test_distributed_gloo.py."""
    from cge.jl_amd import api

    ctx = api.Context(0)
    try:
        ctx.init_rccl(api.rccl_unique_id(), 0, 1)
        rng = np.random.default_rng(0)
        x = rng.standard_normal(100_000)
        assert np.array_equal(ctx.rccl_selftest(x, 0), x) and np.array_equal(ctx.rccl_selftest(x, 1), x)
        k = rng.integers(-2**62, 2**62, 4097)
        assert np.array_equal(ctx.rccl_selftest(k, 2), k)
        assert ctx.get_stat("collective_calls") == 3 and ctx.get_stat("collective_bytes") == 8 * (2 * 100_000 + 4097)
        # a one-rank communicator shards nothing: a score runs as without it
        import cge.jl_amd as cg
        g = os.path.join(GOLDEN, "test115")
        a = cg.parseargs(["-g", f"{g}/test.edgelist", "-c", f"{g}/test1col.ecg", "-e", f"{g}/test_n2v.embedding", "-l", "20",
                          "-f", "1"])
        ctx.set_inputs(a[0], a[1], a[2], a[3], a[5])
        r1 = ctx.score(a[4], 20, 1, "rss", seed=1, auc_samples=500)
        assert np.all(np.isfinite(r1)) and ctx.get_stat("collective_calls") == 3
    finally:
        ctx.close()

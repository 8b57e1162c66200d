"""CPU tests of the host side: parseargs mirror on the reference's three file formats, the C-ABI
library loads and exports every declared symbol, and the product's own host routines (eigenvector,
sampler draw) agree with the oracle / with their contract.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, _parse
from oracle import oracle as orc


def test_parseargs_three_formats_agree():
    # test/runtests.jl:4-19 runs these three argument sets; all describe the same graph + embedding
    g = os.path.join(GOLDEN, "test115")
    a = _parse(["-g", f"{g}/test.edgelist", "-c", f"{g}/test1col.ecg", "-e", f"{g}/test_n2v.embedding", "-l", "20",
                "-f", "1", "-m", "rss"])
    b = _parse(["-g", f"{g}/test.edgelist", "-c", f"{g}/test2col.ecg", "-e", f"{g}/test_ordered.embedding", "-l", "20",
                "-f", "1", "-m", "rss"])
    c = _parse(["-g", f"{g}/test_weights.edgelist", "-c", f"{g}/test2col.ecg", "-e", f"{g}/test_unordered.embedding",
                "-l", "20", "-f", "1", "-m", "rss"])
    for x in (a, b, c):  # test/runtests.jl:21-41
        assert x["edges"].dtype == np.int64 and x["edges"].min() == 1
        assert x["comm"].shape[1] == 1 and x["comm"].min() == 1
        assert x["embedding"].shape == (115, 32) and x["land"] == 20 and x["forced"] == 1
    assert np.array_equal(a["edges"], b["edges"]) and np.array_equal(a["comm"], b["comm"])
    assert np.allclose(a["embedding"], b["embedding"]) and np.allclose(a["embedding"], c["embedding"])
    assert np.allclose(c["eweights"], 1.42) and np.allclose(c["vweights"], 1.42 * a["vweights"])
    assert sorted(map(len, a["clusters"])) == sorted(map(len, b["clusters"]))


def test_parseargs_defaults_and_errors(tmp_path):
    g = os.path.join(GOLDEN, "example10k")
    base = ["-g", f"{g}/10k.edgelist", "-c", f"{g}/10k.ecg", "-e", f"{g}/10k.embedding"]
    auto = _parse(base)  # >= 10 000 vertices: automatic landmarks max(4 sqrt(n), 4 C)  (src/auxilary.jl:194-197)
    assert auto["land"] == max(round(4 * np.sqrt(10000)), 4 * 64) and auto["forced"] == 4
    assert auto["method"].name == "rss" and auto["seed"] == -1 and auto["samples"] == 10000
    exact = _parse(base + ["--force-exact"])
    assert exact["land"] == -1 and exact["clusters"] == {}
    assert _parse(base + ["-l"])["land"] == 400  # -l without a number -> round(4 sqrt(n))  (:176-184)
    # -f without -l sets landmarks = 1 BEFORE the auto-switch test, so the auto-switch never fires (:186-197)
    assert _parse(base + ["-f", "7"])["land"] == 1 and _parse(base + ["-f", "7"])["forced"] == 7
    assert _parse(base + ["-m", " RSS2 ", "-d", "--split-global", "--seed", "3", "--samples-local", "77"])["method"].name == "rss2"
    import cge.jl_amd as cg
    with pytest.raises(cg.ParseError):
        _parse(["-e", f"{g}/10k.embedding"])  # "Edgelist file is required"


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cge_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from cge.jl_amd import api

    lib = api.load_library()
    names = _declared("cge_hip.h") + _declared("cge_hip_testing.h")
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} is declared in include/ but not exported"
    assert lib.cge_abi_version() == 1
    for n in (1, 5, 12):
        for i in range(1, n + 1):
            for j in range(i, n + 1):
                assert api.idx(n, i, j) == orc.idx(n, i, j)


def test_create_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from cge.jl_amd import api

    with pytest.raises(api.CGEError):
        api.Context(0)  # no CPU fallback


def test_host_eigenvector_matches_oracle():
    from cge.jl_amd import api

    lib = api.load_library()
    rng = np.random.default_rng(3)
    for d in (1, 2, 3, 8, 32, 64, 128):
        for trial in range(3):
            k = 3 * d + trial
            y = rng.standard_normal((k, d)) * np.linspace(0.5, 2.0, d)
            A = np.ascontiguousarray(y.T @ y)
            v = np.zeros(d)
            assert lib.cge_host_eig_top(A.ctypes.data_as(C.c_void_p), C.c_int64(d), v.ctypes.data_as(C.c_void_p)) == 0
            ref = orc.eig_top(A)
            assert np.allclose(v, ref, atol=1e-9), (d, np.abs(v - ref).max())
            lam = v @ A @ v
            assert np.linalg.norm(A @ v - lam * v) <= 1e-9 * max(1.0, abs(lam))
    # nearly isotropic covariance (small eigengap): still an eigenvector to working accuracy
    d = 64
    y = rng.standard_normal((2000, d))
    A = np.ascontiguousarray(y.T @ y)
    v = np.zeros(d)
    lib.cge_host_eig_top(A.ctypes.data_as(C.c_void_p), C.c_int64(d), v.ctypes.data_as(C.c_void_p))
    lam = v @ A @ v
    assert np.linalg.norm(A @ v - lam * v) <= 1e-8 * lam
    assert lam == pytest.approx(np.linalg.eigvalsh(A)[-1], rel=1e-12)
    # diagonal matrix (no reflections at all)
    A = np.diag(np.arange(1.0, 9.0))
    v = np.zeros(8)
    lib.cge_host_eig_top(A.ctypes.data_as(C.c_void_p), C.c_int64(8), v.ctypes.data_as(C.c_void_p))
    assert np.allclose(np.abs(v), np.eye(8)[7], atol=1e-12)


def test_sampler_positive_draw_is_uniform_and_reproducible():
    from cge.jl_amd import api

    lib = api.load_library()
    S, m = 200000, 97
    a = np.zeros(S, dtype=np.int64)
    b = np.zeros(S, dtype=np.int64)
    lib.cge_host_pos_draw(C.c_int64(42), C.c_int64(0), C.c_int64(S), C.c_int64(m), a.ctypes.data_as(C.c_void_p))
    lib.cge_host_pos_draw(C.c_int64(42), C.c_int64(0), C.c_int64(S), C.c_int64(m), b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a, b) and a.min() == 1 and a.max() == m
    lib.cge_host_pos_draw(C.c_int64(43), C.c_int64(0), C.c_int64(S), C.c_int64(m), b.ctypes.data_as(C.c_void_p))
    assert not np.array_equal(a, b)
    counts = np.bincount(a, minlength=m + 1)[1:]
    chi2 = ((counts - S / m) ** 2 / (S / m)).sum()
    assert chi2 < 170  # 96 dof: mean 96, sd ~14


def test_text_table_reader_matches_numpy_on_the_reference_files():
    """csrc/textio.cpp (the `readdlm` replacement, src/auxilary.jl:80-168) against numpy on every input file the
    reference ships: same shape, same doubles bit for bit, node2vec header detected."""
    import glob

    from conftest import GOLDEN
    from cge.jl_amd import api

    files = sorted(glob.glob(os.path.join(GOLDEN, "test115", "*"))) + sorted(glob.glob(os.path.join(GOLDEN, "example10k", "10k.*")))
    assert len(files) >= 10
    for f in files:
        got, hdr = api.read_table(f, column_major=False)
        try:
            ref, want_hdr = np.loadtxt(f, ndmin=2), False
        except ValueError:
            ref, want_hdr = np.loadtxt(f, ndmin=2, skiprows=1), True
        assert hdr == want_hdr and got.shape == ref.shape and np.array_equal(got, ref), f
        colmajor, _ = api.read_table(f, column_major=True, n_threads=3)
        assert colmajor.flags["F_CONTIGUOUS"] and np.array_equal(colmajor, ref)


def test_text_table_reader_edge_cases(tmp_path):
    from cge.jl_amd import api

    rng = np.random.default_rng(0)
    # many lines (several pieces per thread), mixed separators, blank lines, CRLF, no trailing newline, signs, exponents
    vals = rng.normal(size=(50000, 7)) * 10.0 ** rng.integers(-12, 12, size=(50000, 7))
    lines = []
    for i, row in enumerate(vals):
        sep = [" ", "\t", "  ", ",", " \t "][i % 5]
        txt = sep.join(("+" if (v > 0 and i % 7 == 0) else "") + repr(float(v)) for v in row)
        lines.append(("  " if i % 11 == 0 else "") + txt + ("  " if i % 13 == 0 else "") + ("\r" if i % 17 == 0 else ""))
        if i % 1000 == 0:
            lines.append("")
    f = tmp_path / "big.txt"
    f.write_text("\n".join(lines))  # no trailing newline
    for nt in (1, 2, 5, 16):
        got, hdr = api.read_table(str(f), column_major=False, n_threads=nt)
        assert not hdr and np.array_equal(got, vals)
    # header line with another field count
    g = tmp_path / "hdr.txt"
    g.write_text("3 2\n1 0.5 1.5\n2 2.5 3.5\n3 4.5 5.5\n")
    got, hdr = api.read_table(str(g), column_major=False)
    assert hdr and np.array_equal(got, [[1, 0.5, 1.5], [2, 2.5, 3.5], [3, 4.5, 5.5]])
    # failures: ragged row, non-numeric field, missing / empty file
    bad = tmp_path / "bad.txt"
    bad.write_text("1 2\n3 4\n5\n6 7\n")
    with pytest.raises(api.CGEError, match="data row 3"):
        api.read_table(str(bad))
    bad.write_text("1 2\n3 x\n")
    with pytest.raises(api.CGEError):
        api.read_table(str(bad))
    with pytest.raises(api.CGEError, match="is not a file"):
        api.read_table(str(tmp_path / "missing.txt"))
    bad.write_text("")
    with pytest.raises(api.CGEError):
        api.read_table(str(bad))


def test_flattened_clusters_are_cached_by_identity():
    """api._flatten_clusters: a list of clusters that was flattened before is recognised by identity (the list and every member
    array) and reused; another list, or the same list with a member REPLACED, is flattened again; FlatClusters passes through."""
    from cge.jl_amd import api

    rng = np.random.default_rng(3)
    cl = [np.sort(rng.choice(5000, size=int(rng.integers(3, 40)), replace=False)) + 1 for _ in range(40)]
    f1, o1 = api._flatten_clusters(cl)
    f2, o2 = api._flatten_clusters(cl)
    assert f1 is f2 and o1 is o2  # the cached arrays themselves
    assert np.array_equal(f1, np.concatenate(cl)) and np.array_equal(np.diff(o1), [len(c) for c in cl])
    cl2 = list(cl)  # another list object: flattened again (same content)
    f3, _ = api._flatten_clusters(cl2)
    assert f3 is not f1 and np.array_equal(f3, f1)
    cl[7] = cl[7][:-1].copy()  # a member replaced: the fingerprint (ids and lengths of the members) no longer matches
    f4, o4 = api._flatten_clusters(cl)
    assert f4 is not f1 and np.array_equal(f4, np.concatenate(cl)) and o4[-1] == len(f1) - 1
    fc = api.flatten_clusters(cl)
    f5, o5 = api._flatten_clusters(fc)
    assert f5 is fc.flat and o5 is fc.off and len(fc) == 40
    small = [np.array([1, 2]), np.array([3])]  # short lists are not cached
    assert api._flatten_clusters(small)[0] is not api._flatten_clusters(small)[0]

import os
import sys

import numpy as np
import pytest

try:  # torch first: it bundles its own HIP runtime, and whichever copy is loaded first serves the whole process -- a test that
    import torch  # noqa: F401  # hands a torch tensor to libcge_hip.so needs both to share one runtime (as in bench.py)
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _parse(argv):
    import cge.jl_amd as cg

    keys = ("edges", "eweights", "vweights", "comm", "clusters", "embedding", "verbose", "land", "forced", "method",
            "directed", "split", "seed", "samples")
    return dict(zip(keys, cg.parseargs(argv, exit_on_error=False)))


@pytest.fixture(scope="session")
def test115():
    """The reference's own test fixture (test/runtests.jl:4-7): 115 vertices, 613 edges, d = 32."""
    g = os.path.join(GOLDEN, "test115")
    return _parse(["-g", f"{g}/test.edgelist", "-c", f"{g}/test1col.ecg", "-e", f"{g}/test_n2v.embedding", "-l", "20",
                   "-f", "1", "-m", "rss"])


@pytest.fixture(scope="session")
def example10k():
    """example/10k.* with the README flags (README.md:88-100)."""
    g = os.path.join(GOLDEN, "example10k")
    return _parse(["-g", f"{g}/10k.edgelist", "-c", f"{g}/10k.ecg", "-e", f"{g}/10k.embedding", "-l", "200", "--seed",
                   "42"])


def random_samples(rng, m, n, S, n_sets=1):
    pos = rng.integers(1, m + 1, size=(n_sets, S))
    ni = rng.integers(1, n + 1, size=(n_sets, S))
    nj = rng.integers(1, n + 1, size=(n_sets, S))
    nj[ni == nj] = (nj[ni == nj] % n) + 1
    return pos, ni, nj


def canonical_partition(labels):
    """Relabel a partition by order of first appearance (ids independent of numbering)."""
    labels = np.asarray(labels)
    _, first = np.unique(labels, return_index=True)
    order = np.argsort(first)
    remap = np.empty(len(order), dtype=np.int64)
    remap[order] = np.arange(len(order))
    _, inv = np.unique(labels, return_inverse=True)
    return remap[inv]

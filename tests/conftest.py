import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_libsvm(path):
    """label idx:val ... ; indices kept as-is (1-based ids stay 1-based)"""
    off, idx, val, lab = [0], [], [], []
    with open(path) as f:
        for line in f:
            t = line.split()
            if not t:
                continue
            lab.append(float(t[0]))
            for kv in t[1:]:
                k, v = kv.split(":")
                idx.append(int(k))
                val.append(float(v))
            off.append(len(idx))
    return (np.array(off, np.uint64), np.array(idx, np.uint64), np.array(val, np.float32),
            np.array(lab, np.float32))


@pytest.fixture(scope="session")
def rcv1():
    """the reference's own fixture tests/data: first 100 rows of rcv1.binary"""
    off, idx, val, lab = load_libsvm(os.path.join(GOLDEN, "rcv1_100.libsvm"))
    return dict(offset=off, index=idx, value=val, label=lab)


@pytest.fixture(scope="session")
def oracle():
    from oracle import bindings
    bindings.build(ref=os.path.isdir("/root/reference/src"))
    return bindings.Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle import bindings
    if not bindings.have_ref():
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    return bindings.Ref()


def random_batch(rng, nrows, nfeat_space, max_nnz_row, binary=False, empty_rows=True):
    """a ragged random CSR batch with raw u64 ids"""
    lens = rng.integers(0 if empty_rows else 1, max_nnz_row + 1, size=nrows)
    off = np.zeros(nrows + 1, np.uint64)
    off[1:] = np.cumsum(lens)
    nnz = int(off[-1])
    idx = rng.integers(0, nfeat_space, size=nnz, dtype=np.uint64)
    val = None if binary else rng.normal(size=nnz).astype(np.float32)
    lab = np.where(rng.random(nrows) < 0.4, 1.0, -1.0).astype(np.float32)
    return dict(offset=off, index=idx, value=val, label=lab)

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available():
    try:
        from point_cloud_utils_amd import _lib
        return _lib.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip: the product has no CPU fallback.
    pass


@pytest.fixture(scope="session")
def oracle_kind():
    import oracle
    oracle.build()
    return "ref" if oracle.have_ref() else "port"


def cloud(seed, n, dtype, scale=1.0, offset=0.0):
    """SURVEY 8d synthetic input: i.i.d. U[0,1)^3, C-contiguous."""
    rng = np.random.default_rng(seed)
    a = rng.random((n, 3), dtype=dtype)
    if scale != 1.0 or offset != 0.0:
        a = (a * dtype(scale) + dtype(offset)).astype(dtype)
    return a


def read_ply_vertices(path):
    """Minimal binary-little-endian PLY vertex reader (x,y,z float/double first in the vertex element)."""
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt = None; n = 0; props = []; in_vertex = False
        while True:
            line = f.readline().strip()
            if line == b"end_header":
                break
            tok = line.split()
            if tok[0] == b"format": fmt = tok[1]
            elif tok[0] == b"element":
                in_vertex = tok[1] == b"vertex"
                if in_vertex: n = int(tok[2])
            elif tok[0] == b"property" and in_vertex:
                props.append((tok[2].decode(), tok[1].decode()))
        assert fmt == b"binary_little_endian"
        m = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "int": "<i4", "uint": "<u4"}
        dt = np.dtype([(nm, m[t]) for nm, t in props])
        v = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    return np.stack([v["x"], v["y"], v["z"]], axis=-1)

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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


def mesh_samples(v, f, n, seed=5):
    """BASELINE config 5: n area-weighted uniform samples on the triangles (v, f) -- face ~ areas, barycentric
    (1 - sqrt(u), sqrt(u) (1 - w), sqrt(u) w) -- in float64, C-contiguous."""
    rng = np.random.default_rng(seed)
    tri = v[f]
    areas = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    fi = rng.choice(len(f), n, p=areas / areas.sum())
    u = rng.random(n); w = rng.random(n); su = np.sqrt(u)
    s = (1 - su)[:, None] * tri[fi, 0] + (su * (1 - w))[:, None] * tri[fi, 1] + (su * w)[:, None] * tri[fi, 2]
    return np.ascontiguousarray(s)


def read_ply_vertices(path):
    """Minimal binary-little-endian PLY reader returning the (n,3) x,y,z of the `vertex` element. Elements may
    come in any order (data/bunny_duplicates.ply stores faces first); list properties are walked to skip them."""
    import struct
    m = {"char": "b", "uchar": "B", "short": "h", "ushort": "H", "int": "i", "uint": "I", "float": "f", "double": "d",
         "int8": "b", "uint8": "B", "int16": "h", "uint16": "H", "int32": "i", "uint32": "I", "float32": "f", "float64": "d"}
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt = None; elements = []
        while True:
            line = f.readline().strip()
            if line == b"end_header":
                break
            tok = line.split()
            if tok[0] == b"format": fmt = tok[1]
            elif tok[0] == b"element": elements.append([tok[1].decode(), int(tok[2]), []])
            elif tok[0] == b"property":
                if tok[1] == b"list": elements[-1][2].append(("list", tok[2].decode(), tok[3].decode(), tok[4].decode()))
                else: elements[-1][2].append((tok[2].decode(), tok[1].decode()))
        assert fmt == b"binary_little_endian"
        for name, count, props in elements:
            has_list = any(p[0] == "list" for p in props)
            if name == "vertex":
                assert not has_list
                dt = np.dtype([(nm, "<" + m[t]) for nm, t in props])
                v = np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)
                return np.stack([v["x"], v["y"], v["z"]], axis=-1)
            if not has_list:
                f.seek(count * sum(struct.calcsize(m[t]) for _, t in props), 1)
            else:
                for _ in range(count):
                    for p in props:
                        if p[0] == "list":
                            (c,) = struct.unpack("<" + m[p[1]], f.read(struct.calcsize(m[p[1]])))
                            f.seek(c * struct.calcsize(m[p[2]]), 1)
                        else:
                            f.seek(struct.calcsize(m[p[1]]), 1)
    raise ValueError("no vertex element")

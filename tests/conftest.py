import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    if name == "cat_mesh.npz":              # the benchmark mesh is package data (bench.py needs it too): raytracinggpu_amd/data/
        return np.load(os.path.join(ROOT, "raytracinggpu_amd", "data", name), allow_pickle=False)
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def cat_golden():
    return load_golden("cat_mesh.npz")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def oracle_cat(oracle, cat_golden):
    """scene_cpu mesh: the reference parser's vertices/indices (OBJ order) + the oracle's own BVH build."""
    return oracle.Mesh.from_arrays(cat_golden["vertices"], cat_golden["tri_obj_order"]).build_bvh()

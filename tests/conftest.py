import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP device)")


class Golden:
    def __init__(self):
        self.arrays = np.load(os.path.join(GOLDEN_DIR, "golden.npz"))
        with open(os.path.join(GOLDEN_DIR, "golden.json")) as f:
            self.manifest = json.load(f)

    def __getitem__(self, key):
        return self.arrays[key]

    def x_of(self, case):
        if "x" in case:
            return self.arrays[case["x"]]
        return gen_input(case["x_gen"])

    def cases(self, kind):
        return self.manifest[kind]


def gen_input(spec):
    rng = np.random.default_rng(spec["seed"])
    if spec["dist"] == "uniform":
        return rng.random(tuple(spec["shape"]))
    return rng.standard_normal(tuple(spec["shape"]))


_GOLDEN = None


def load_golden():
    global _GOLDEN
    if _GOLDEN is None:
        _GOLDEN = Golden()
    return _GOLDEN


@pytest.fixture(scope="session")
def golden():
    return load_golden()

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


# Parity report: every feature comparison of the GPU suite records how many counting
# features were compared, how many were exposed to a threshold tie, and how many differed;
# printed at the end of the run (pytest -m gpu) so the observed mismatches are on record.
PARITY_REPORT = []


def pytest_terminal_summary(terminalreporter):
    if not PARITY_REPORT:
        return
    tr = terminalreporter
    tr.write_sep("-", "feature parity report (counting sieves vs the oracle)")
    tot = {"entries": 0, "exposed": 0, "differ": 0, "differ_unexposed": 0}
    for r in PARITY_REPORT:
        for k in tot:
            tot[k] += r[k]
    worst = sorted(PARITY_REPORT, key=lambda r: -r["differ"])[:12]
    for r in worst:
        if r["differ"] == 0:
            break
        tr.write_line(f"  {r['what'][:70]:70s} entries {r['entries']:8d} exposed {r['exposed']:7d} "
                      f"differ {r['differ']:5d} (max |d| {r['max_d']:.0f}, unexposed {r['differ_unexposed']})")
    tr.write_line(f"  TOTAL over {len(PARITY_REPORT)} comparisons: count entries {tot['entries']}, "
                  f"tie-exposed {tot['exposed']}, differing {tot['differ']} "
                  f"({100.0 * tot['differ'] / max(tot['entries'], 1):.4f} %), "
                  f"differing outside exposure {tot['differ_unexposed']}")

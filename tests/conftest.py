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
    # Fruit.transform's unasked-for preparation: what the disk cache and the kernels shipped with
    # the build hold, but no background compilations (the tests that are about those set the
    # variable themselves; tools/gpu_knobs.sh runs the suite with =all) - a dozen full-size
    # transforms would otherwise each start seconds of compiler that the interpreter joins at exit
    os.environ.setdefault("FRUITS_AMD_AUTO_PREPARE", "cached")


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
# The same for the STRICT comparisons (the oracle's thresholds transplanted into the GPU
# pipeline, tight exposure) and for the fit comparisons (thresholds fitted here vs the oracle's).
STRICT_REPORT = []
FIT_REPORT = []


def _totals(tr, title, report):
    tr.write_sep("-", title)
    tot = {"entries": 0, "exposed": 0, "differ": 0, "differ_unexposed": 0}
    for r in report:
        for k in tot:
            tot[k] += r[k]
    worst = sorted(report, key=lambda r: -r["differ"])[:12]
    for r in worst:
        if r["differ"] == 0:
            break
        tr.write_line(f"  {r['what'][:70]:70s} entries {r['entries']:8d} exposed {r['exposed']:7d} "
                      f"differ {r['differ']:5d} (max |d| {r['max_d']:.0f}, unexposed {r['differ_unexposed']})")
    tr.write_line(f"  TOTAL over {len(report)} comparisons: count entries {tot['entries']}, "
                  f"tie-exposed {tot['exposed']} ({100.0 * tot['exposed'] / max(tot['entries'], 1):.3f} %), "
                  f"differing {tot['differ']} "
                  f"({100.0 * tot['differ'] / max(tot['entries'], 1):.4f} %), "
                  f"differing outside exposure {tot['differ_unexposed']}")
    return tot


def pytest_terminal_summary(terminalreporter):
    tr = terminalreporter
    if PARITY_REPORT:
        _totals(tr, "feature parity report, end to end (fit AND transform on the GPU vs the oracle; "
                    "exposure 1e-10 of the row)", PARITY_REPORT)
    for cls, title in (("sum", "columns of sums (Reals, CosWISS)"),
                       ("max-plus", "columns of max-plus slices (Arctic / Bayesian: plateaus)")):
        part = [r for r in STRICT_REPORT if r.get("cls") == cls]
        if not part:
            continue
        tot = _totals(tr, f"feature parity report, transform alone - {title}: the oracle's thresholds in the "
                          "GPU pipeline; exposure 1e-13 of the differenced row + 8 ulp of the row", part)
        means = sum(r.get("means_checked", 0) for r in part)
        tr.write_line(f"  band means of exposed entries checked against the candidate means: {means}")
        wide = [r for r in part if r.get("series", 0) >= 64]
        if wide:
            we, wx = sum(r["entries"] for r in wide), sum(r["exposed"] for r in wide)
            tr.write_line(f"  on batches of >= 64 series: {we} entries, {wx} exposed ({100.0 * wx / max(we, 1):.3f} %)")
        big = max((r["max_d"] for r in part), default=0)
        tr.write_line(f"  SURVEY 7 bar: differing {100.0 * tot['differ'] / max(tot['entries'], 1):.4f} % "
                      f"(bar 0.1 %), largest |d| {big:.0f} (bar 1)"
                      + ("" if cls == "sum" else f" - this class must be EQUAL: {cls}: differing {tot['differ']}"))
    if FIT_REPORT:
        tr.write_sep("-", "fit parity report (thresholds fitted on the GPU vs the oracle's)")
        n = sum(r["thresholds"] for r in FIT_REPORT)
        same = sum(r["identical"] for r in FIT_REPORT)
        worst = max((r["max_rel"] for r in FIT_REPORT), default=0.0)
        tr.write_line(f"  {n} finite thresholds over {len(FIT_REPORT)} fits: {same} bit-identical, largest "
                      f"deviation {worst:.2e} of the magnitude of the fitted rows (bar 1e-13)")

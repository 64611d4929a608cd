"""pytest configuration: the ``gpu`` marker and shared fixtures.

``-m "not gpu"`` : oracle vs golden vectors, host logic, C-ABI symbol export (no compute calls).
``-m gpu``       : parity tests proper - the HIP path through the C-ABI vs oracle / golden vectors.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The CPU oracle is the slow half of every parity test.  A GPU box gives a 1-GPU job 16 cores of a 256-thread host, and
# torch would start one thread per core it SEES: bench.py's probe of the oracle there reads {8: 0.34 s, 16: 0.29 s,
# 64: 1.1 s}.  tests/reports/parity_report.py imports this module too, so the recorded oracle hashes are taken at the
# same thread count as the tests' own oracle runs.
torch.set_num_threads(min(16, os.cpu_count() or 1))


# The frozen MPJPE bounds (tests/parity_bounds.json): the SHA-256 of its `cases` object (keys sorted, no whitespace).  A changed bound
# fails the whole session here, so loosening one is a visible diff in TWO files (the file was rewritten in the round
# that froze it).  The `what` text is not part of the digest.
PARITY_BOUNDS_CASES_SHA256 = "f64ef3ce6dc32b59109908d46f7b2f04e78f9c79ebed44f79810ddfb0c1d3898"


def _check_parity_bounds_digest():
    import hashlib
    import json
    with open(os.path.join(ROOT, "tests", "parity_bounds.json")) as f:
        cases = json.load(f)["cases"]
    got = hashlib.sha256(json.dumps(cases, sort_keys=True, separators=(",", ":")).encode()).hexdigest()
    if got != PARITY_BOUNDS_CASES_SHA256:
        raise pytest.UsageError(f"tests/parity_bounds.json: the bounds changed (sha256 of `cases` {got}, pinned {PARITY_BOUNDS_CASES_SHA256}); "
                                "frozen bounds are not edited - if a bound must move, change the pin in tests/conftest.py in the same commit and say why")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _check_parity_bounds_digest()


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: torch.from_numpy(z[k]) for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_terminal_summary(terminalreporter):
    """the parity cases' measured |dMPJPE| against their bounds and against north_star's 1e-4 mm (tests/test_hip_parity.py)"""
    try:
        from tests.test_hip_parity import PARITY_LINES
    except Exception:
        return
    if PARITY_LINES:
        terminalreporter.write_sep("-", "MPJPE parity (HIP path vs CPU oracle)")
        for line in PARITY_LINES:
            terminalreporter.write_line(line)

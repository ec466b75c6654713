import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


SUITE_BUDGET_S = 1200.0      # the driver's limit for `pytest -m gpu`


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes >= 10 s on the GPU box (still part of -m gpu; deselect with -m 'gpu and not slow')")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Always print where the time went (the GPU suite runs under a 1200 s limit): total and the ten slowest test calls,
    with a warning once the suite has used more than two thirds of the budget."""
    calls = []
    for reports in terminalreporter.stats.values():
        for r in reports:
            if getattr(r, "when", None) == "call" and hasattr(r, "duration"):
                calls.append((r.duration, r.nodeid))
    if not calls:
        return
    total = sum(d for d, _ in calls)
    terminalreporter.write_sep("-", f"test time: {total:.0f} s in {len(calls)} calls (GPU suite budget {SUITE_BUDGET_S:.0f} s)")
    for d, n in sorted(calls, reverse=True)[:10]:
        terminalreporter.write_line(f"{d:8.2f} s  {n}")
    if total > SUITE_BUDGET_S * 2 / 3:
        terminalreporter.write_line(f"WARNING: the suite uses {total / SUITE_BUDGET_S:.0%} of its time budget; mark or trim the slow tests")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must FAIL (not skip) on a box without a GPU or without the HIP library:
    a silent skip would hide a missing native path.  Without `-m gpu` they are deselected by
    the marker expression the driver passes (-m "not gpu")."""
    return


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session", autouse=True)
def configs():
    """configs/ is generated, not committed: tools/emit_configs.py writes the task stacks, CONTROLLER / BEHAVIOR trees and model
    files the C++ facade reads from the constants in inria_wbc_amd/ (the directory is git-ignored)."""
    from tools import emit_configs
    emit_configs.main()
    return os.path.join(ROOT, "configs")


@pytest.fixture(scope="session")
def built_lib():
    """libwbcqp.so, (re)built in-tree with hipcc when sources changed (cross-compiles without a GPU)."""
    from inria_wbc_amd import build
    return build.build()


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def pytest_terminal_summary(terminalreporter):
    """The figures tests/util.py:assert_parity measured in this session, one line per case: the bars bound them, this prints them (raw contact-point
    forces in particular: bounded at 1e-4, reported here; active sets: share of QPs whose set equals the oracle's bit for bit)."""
    from tests import util
    if not util.MEASURED:
        return
    tr = terminalreporter
    tr.section("parity figures measured against the oracle (tests/util.py:assert_parity)")
    for what, info in util.MEASURED:
        tr.write_line("%-58s dv %.1e  wrench %.1e  tau %.1e  raw f %.1e  iters= %.3f  active set= %s (%s checked, facets differ on %s)  objective %s" % (
            what[:58], info.get("max_rel_dv", 0.0), info.get("max_rel_wrench", 0.0), info.get("max_rel_tau", 0.0), info.get("max_rel_raw_force", 0.0),
            info.get("iters_equal", 1.0), ("%.3f" % info["active_set_equal_frac"]) if "active_set_equal_frac" in info else "-",
            info.get("active_set_checked", "-"), info.get("facets_differ", "-"), ("%.1e" % info["max_rel_objective"]) if "max_rel_objective" in info else "-"))

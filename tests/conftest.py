import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter):
    """Worst pass margin of every increment-parity check (tests/util.py): ratio 1.0 = at the bar."""
    try:
        from tests.util import WORST
    except Exception:
        return
    if WORST:
        terminalreporter.write_line("increment-parity margins (|d_gpu - d_oracle| / (1e-4 |d_oracle| + k ulp32(M)); <= 1 passes):")
        for k, v in sorted(WORST.items()):
            terminalreporter.write_line(f"  {k:<60s} {v:.3f}")
        out = os.environ.get("DSIM_MARGINS_OUT")
        if out:
            import json
            from tests.util import WHERE
            with open(out, "w") as fh:
                json.dump({k: round(v, 4) for k, v in sorted(WORST.items())}, fh, indent=1)
            with open(os.path.splitext(out)[0] + "_where.json", "w") as fh:
                json.dump({k: v for k, v in sorted(WHERE.items())}, fh, indent=1)

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """Builds the checker libraries: the CPU restatement always, oracle/_ref only where the
    reference sources exist (this container); on the GPU box the prebuilt _ref .so travels."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle", "ref"])


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))

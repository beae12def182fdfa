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


@pytest.fixture(scope="session", autouse=True)
def _debug_switches():
    """Scorers created by the tests take the route switches the tests name in the environment (tests/switches.py);
    the library itself reads no route from the environment."""
    import switches
    switches.install()


def checker_kind():
    """Which CPU checker the parity tests compare the HIP path with.  "ref" = the reference's own
    C++ core (oracle/_ref/libascore_ref.so: built here from /root/reference, prebuilt on the GPU
    box).  Its absence is an error, not a silent switch to this repo's own restatement: set
    PYA_ALLOW_PORT_CHECKER=1 to run against oracle/ascore_oracle.cpp knowingly (the kind is
    printed in the pytest header either way)."""
    from oracle import orc
    if orc.available("ref"):
        return "ref"
    if os.environ.get("PYA_ALLOW_PORT_CHECKER"):
        return "oracle"
    pytest.fail("oracle/_ref/libascore_ref.so is missing: the GPU parity tests compare against the "
                "reference's own C++ core.  Build it with `make -C oracle ref` where /root/reference exists, "
                "or set PYA_ALLOW_PORT_CHECKER=1 to use this repo's CPU restatement instead.", pytrace=False)


def pytest_report_header(config):
    from oracle import orc
    kind = "ref (reference C++ core, oracle/_ref)" if orc.available("ref") else (
        "oracle (this repo's restatement; PYA_ALLOW_PORT_CHECKER set)" if os.environ.get("PYA_ALLOW_PORT_CHECKER")
        else "MISSING oracle/_ref -- GPU parity tests will fail")
    return "pyascore_amd parity checker: " + kind


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))

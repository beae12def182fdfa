"""The pin of the pin: tests/golden/*.npz are generated through oracle/_ref (the reference's C++ core behind a shim);
this replays every one of them through the reference's REAL Cython extension, built by the reference's own setup.py
under /tmp (tests/golden/verify_against_cython.py), and requires bit equality.  Runs only where /root/reference
exists (the build container); the GPU box has the committed vectors and nothing of the reference."""
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN, golden_cases


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference lives in the build container only")
def test_goldens_are_bit_equal_to_the_reference_cython_extension():
    p = subprocess.run([sys.executable, os.path.join(GOLDEN, "verify_against_cython.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-4000:]
    last = p.stdout.strip().splitlines()[-1]
    assert "ALL BIT-EQUAL" in last and ("%d golden files" % len(golden_cases())) in last, last

"""charge_mz (csrc/device_common.hip.h) divides by the fragment charge with five fused multiply-adds instead of the f64 divide
sequence; the quotient must be the correctly rounded one (the reference divides: cpp/ModifiedPeptide.cpp:586-588).  The
sequence is the same arithmetic on any IEEE machine with an FMA: scripts/divcheck.c runs it on the host against the
division, for every charge 3 .. 255 that is not a power of two, on random operands."""
import os
import subprocess

import pytest

from conftest import ROOT


def test_fma_sequence_equals_the_division(tmp_path):
    src = os.path.join(ROOT, "scripts", "divcheck.c")
    exe = str(tmp_path / "divcheck")
    flags = open("/proc/cpuinfo").read()
    if " fma" not in flags:
        pytest.skip("this CPU has no FMA instruction (a software fma() would be exact too, but slow)")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-DITERS=100000", src, "-o", exe, "-lm"])
    out = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "two corrections differ 0" in out.stdout and "one correction differs 0" in out.stdout, out.stdout

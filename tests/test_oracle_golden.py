"""CPU restatement (oracle/ascore_oracle.cpp) against the committed golden vectors, which were
produced by the reference's own C++ core (tests/golden/make_golden.py).  Bit-exact."""
import os

import pytest

from conftest import GOLDEN, golden_cases
from oracle import harness, orc
from pyascore_amd import synth


@pytest.mark.parametrize("case", golden_cases())
def test_oracle_matches_golden(case):
    settings, batch, expected = harness.load_case(os.path.join(GOLDEN, case + ".npz"))
    scorer = harness.make_scorer(orc.OracleAscore, settings, kind="oracle")
    got = harness.collect(scorer, batch, synth.unpack_psm)
    assert harness.compare(got, expected, exact_float=True) == []


@pytest.mark.parametrize("case", ["velos_z1", "synth_cfg3", "edge_nl"])
def test_oracle_batch_driver_matches_golden(case):
    """orc_score_batch (the timed CPU-baseline entry) returns the same summary."""
    import numpy as np
    settings, batch, expected = harness.load_case(os.path.join(GOLDEN, case + ".npz"))
    scorer = harness.make_scorer(orc.OracleAscore, settings, kind="oracle")
    got = scorer.score_batch(batch, max_k=expected["ascores"].shape[1])
    for k in ("best_score", "best_sig", "n_sig", "ascores", "alt_mask"):
        assert np.array_equal(got[k], expected[k]), k

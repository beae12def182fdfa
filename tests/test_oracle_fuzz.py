"""CPU restatement vs the reference's own C++ core on random settings and batches (same draws as
tests/test_gpu_fuzz.py).  Skipped where oracle/_ref is absent."""
import numpy as np
import pytest

from fuzzcase import random_case
from oracle import harness, orc

pytestmark = pytest.mark.skipif(not orc.available("ref"), reason="oracle/_ref not built")


@pytest.mark.parametrize("seed", range(40))
def test_oracle_equals_reference_on_random_cases(seed):
    settings, batch = random_case(np.random.default_rng(9000 + seed))
    if batch["n_psm"] == 0:
        pytest.skip("empty draw")
    k = max(1, int(batch["n_of_mod"].max()))
    a = harness.make_scorer(orc.OracleAscore, settings, kind="ref").score_batch(batch, k)
    b = harness.make_scorer(orc.OracleAscore, settings, kind="oracle").score_batch(batch, k)
    for key in a:
        assert np.array_equal(a[key], b[key]), key

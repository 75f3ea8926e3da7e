"""A fixed set of seeds of the randomised differential test (fuzz_parity.py): random scenes, trees (median and SAH-binned),
cameras (also inside the model), feature combinations, sizes, start depths, flags and frame counts, HIP path against the oracle."""
import pytest

import fuzz_parity as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("first", [0, 40, 80])
def test_random_cases_match_the_oracle(first):
    failures = []
    for seed in range(first, first + 40):
        c = F.draw_case(seed)
        bad = F.run_case(c)
        if bad:
            failures.append(F.describe(c) + " -> " + "; ".join(bad))
    assert not failures, "\n".join(failures)

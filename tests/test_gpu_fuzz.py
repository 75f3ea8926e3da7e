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


@pytest.mark.parametrize("first", [0, 60])
def test_tie_heavy_generated_scenes_match_the_oracle(first):
    """Generated scenes whose triangles lie on a coarse grid, some stored twice: exact ties of distance, rays through shared edges and
    vertices, and - under the SAH-binned build - leaves of several hundred triangles."""
    failures, biggest = [], 0
    for seed in range(first, first + 60):
        c = F.draw_case(seed, soups=True)
        bad = F.run_case(c)
        g, _ = F.scene_pair(c["scene"], c["sah"], c["transparent"])
        nodes = g.bvh.nodes()
        biggest = max(biggest, int(nodes[(nodes[:, 6] >> 31) == 1][:, 7].max()))
        if bad:
            failures.append(F.describe(c) + " -> " + "; ".join(bad))
    assert not failures, "\n".join(failures)
    assert biggest > 63  # leaves beyond the count field of a leaf reference were among them

"""The OBJ loader behind bench.py --obj: structure on a synthetic file, and - where the reference's data
directory is present (the build container only) - bit-identity with the reference loader's output that
tests/golden/scenes.npz holds."""
import os

import numpy as np
import pytest

from raytracer_amd import scenes

REF_DATA = "/root/reference/data"


def test_synthetic_obj(tmp_path):
    p = tmp_path / "m.obj"
    p.write_text("""
v 0 0 0
v 2 0 0
v 2 1 0
v 0 1 0
v 0 0 3
vn 0 0 1
f 1 2 3 4
f 1//1 2//1 5//1
""")
    s = scenes.load_obj(str(p), normalize=False)
    f = s["prims"].view(np.float32).reshape(-1, 25)
    assert len(f) == 3 and s["materials"].tolist() == [[1, 1, 1, 0, 0, 0, 1, 1, 0]]
    # the 2x1 quad has equal diagonals: tinyobjloader then takes [0,1,3],[1,2,3]
    assert f[0, 1:4].tolist() == [0, 0, 0] and f[0, 9:12].tolist() == [2, 0, 0] and f[0, 17:20].tolist() == [0, 1, 0]
    assert f[1, 1:4].tolist() == [2, 0, 0] and f[1, 17:20].tolist() == [0, 1, 0]
    assert f[0, 4:7].tolist() == [0, 0, 1]  # geometric normal where the file gives none
    assert f[2, 4:7].tolist() == [0, 0, 1]  # the file's normal where it gives one
    n = scenes.load_obj(str(p)).get("prims").view(np.float32).reshape(-1, 25)
    pos = np.concatenate([n[:, 1:4], n[:, 9:12], n[:, 17:20]])
    assert abs(np.linalg.norm(pos, axis=1).max() - 1.0) < 1e-6  # scaled to the unit ball around the vertex mean


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference data directory not present")
@pytest.mark.parametrize("name,file", [("monkey", "monkey.obj"), ("cornell", "CornellBox-Mirror-Rotated.obj"), ("teapot", "teapot.obj")])
def test_loader_reproduces_reference_loader(name, file):
    got = scenes.load_obj(os.path.join(REF_DATA, file))
    want = scenes.preset(name)
    assert np.array_equal(got["prims"], want["prims"])
    assert np.array_equal(got["materials"], want["materials"])

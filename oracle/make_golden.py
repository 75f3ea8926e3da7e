#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: regenerate tests/golden/*.npz from the reference itself.

Builds oracle/_ref/ref_harness (reference sources compiled where they lie under
/root/reference, see oracle/Makefile and oracle/ref/ref_harness.cpp), runs it, and packs the
.npy arrays it writes into two compressed archives:

  tests/golden/ref_vectors.npz  known-answer vectors for sampler / interpolation / shading /
                                light sampling / reflection rays / glossy helpers / bloom filter
  tests/golden/scenes.npz       the reference loader's output for its preset scenes
                                (primitives in BVH gather order, materials, lights, spheres)

Only runs where /root/reference exists (the build container); the GPU box uses the committed
archives.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main() -> int:
    ref = os.environ.get("REF", "/root/reference")
    if not os.path.isdir(ref):
        print(f"{ref} not present; keeping the committed golden archives")
        return 0
    subprocess.check_call(["make", "-C", HERE, "-j8", "ref", f"REF={ref}"])
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([os.path.join(HERE, "_ref", "ref_harness"), tmp])
        vec, scn = {}, {}
        for fn in sorted(os.listdir(tmp)):
            if not fn.endswith(".npy"):
                continue
            arr = np.load(os.path.join(tmp, fn))
            (scn if fn.startswith("scene_") else vec)[fn[:-4]] = arr
    np.savez_compressed(os.path.join(out_dir, "ref_vectors.npz"), **vec)
    np.savez_compressed(os.path.join(out_dir, "scenes.npz"), **scn)
    for name in ("ref_vectors.npz", "scenes.npz"):
        p = os.path.join(out_dir, name)
        print(f"{p}: {os.path.getsize(p)} bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())

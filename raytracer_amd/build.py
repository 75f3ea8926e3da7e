"""Builds raytracer_amd/lib/librt_amd.so (HIP kernels + C ABI) for gfx950 with hipcc.

The flags are part of the numerics contract: no FMA contraction, no fast-math, correctly rounded
fp32 divide/sqrt, denormals kept - float ties decide which triangle a ray reports."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "librt_amd.so")
SOURCES = ["rt_device.hip", "query_build.hip", "ref_build.hip", "host_bvh.cpp", "rt_multi.cpp"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "rt_amd.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-Wall", "-Wno-unused-function"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False, extra=()) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + ["-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, extra=[a for a in sys.argv[1:] if a.startswith(("-R", "-save", "-D"))])
    print(LIB)

"""Wall time of rt_bvh_build, rt_bvh_build_device and rt_scene_create for a bench scene, several times in one process (the first pays for loading the
library's kernels).  usage: scene_create_probe.py [scene] [repeats];  RT_AMD_VERBOSE=1 prints the library's own breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from raytracer_amd import api
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = bench.build_scene_arrays(name)
for k in range(n):
    t0 = time.perf_counter()
    bvh = api.Bvh(sc["prims"], False)
    host_ms = (time.perf_counter() - t0) * 1e3
    for sah in (False, True):
        ta = time.perf_counter()
        dev = api.Bvh(sc["prims"], sah, device=True)
        tb = time.perf_counter()
        print("%s: rt_bvh_build_device(%s) %.1f ms %s" % (name, "sah" if sah else "median", (tb - ta) * 1e3, dev.device_info), flush=True)
        del dev
    t1b = time.perf_counter()
    hs = api.Bvh(sc["prims"], True)
    print("%s: rt_bvh_build_sah_binning %.1f ms" % (name, (time.perf_counter() - t1b) * 1e3), flush=True)
    del hs
    t1 = time.perf_counter()
    s = api.Scene(bvh.nodes(), bvh.primitives(), bvh.primitive_ids(), sc["materials"], sc["lights"], sc["spheres"])
    t2 = time.perf_counter()
    print("%s: rt_bvh_build %.1f ms, rt_scene_create %.1f ms (incl. the binding's array copies)" % (name, host_ms, (t2 - t1) * 1e3), flush=True)
    del s

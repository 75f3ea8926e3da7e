import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_AMD_DBG"] = "/tmp/dbg.bin"
sys.argv = ["frame.py", "2"]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "frame.py")).read())
d = np.fromfile("/tmp/dbg.bin", np.uint32).reshape(-1, 3)
d = d[d[:, 1] > 0]
t = d[:, 0] / 100.0
print("waves with work:", len(d), "wave lifetime us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (t.mean(), *np.percentile(t, [50, 90, 99]), t.max()))
print("iterations per wave: mean %.0f max %d; rays per wave mean %.0f; us per iteration %.2f" % (d[:, 1].mean(), d[:, 1].max(), d[:, 2].mean() , (t.sum() / d[:, 1].sum())))

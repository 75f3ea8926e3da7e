"""Looks closer at seeds of fuzz_parity.py that exceed the bar: is the difference one of the two the reference's own algorithm is
discontinuous in - a glossy sample rejected or kept on `dot(direction, normal) <= 0` (src/extra.cpp:497), a pixel inside or
outside the bloom's threshold (src/extra.cpp:330-416) - where the GPU's sinf / cosf / powf differ from glibc's in the last ulp?

    python tests/fuzz_diagnose.py [big|soup] <seed>... [feature=value ...]     (overrides applied to every seed's features)
"""
import sys

import numpy as np

import fuzz_parity as F
from fuzz_parity import O, api


def main():
    override = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
    mode = [a for a in sys.argv[1:] if a in ("big", "soup")]
    for seed in [int(a) for a in sys.argv[1:] if a.isdigit()]:
        c = F.draw_case(seed, "big" in mode, "soup" in mode)
        c["features"].update(override)
        c["frames"] = min(c["frames"], 2)
        print(F.describe(c))
        g, o = F.scene_pair(c["scene"], c["sah"], c["transparent"])
        W, H = c["W"], c["H"]
        f = dict(c["features"])
        bloom = f.pop("enableBloomEffect", 0)
        fg, fo = api.Features(**f), O.make_features(**f)
        for k in range(c["frames"]):
            kw = dict(c["cam"])
            kw["rot_deg"] = (kw["rot_deg"][0], kw["rot_deg"][1] + 37.0 * k, 0.0)
            cg, co = api.camera_from_trackball(W, H, **kw), O.make_camera(W, H, **kw)
            O.lib().orc_set_shadow_walk_cap(c.get("walk_cap", 0))  # (a case with light through transparent surfaces bounds the walk on both sides)
            try:
                ro = o.render(co, fo, W, H, start_depth=c["start_depth"])
            finally:
                O.lib().orc_set_shadow_walk_cap(0)
            rays_o = (ro["stats"]["primary"], ro["stats"]["shadow"], ro["stats"]["secondary"])
            frames = {}
            for name, flags in (("as drawn", c["flags"]), ("depth-first", api.RT_RENDER_SERIAL), ("caller's tree", api.RT_RENDER_REFERENCE_TREE)):
                r = g.render_image(cg, fg, W, H, start_depth=c["start_depth"], flags=flags)
                frames[name] = r
                d = np.abs(r["rgb"].astype(np.float64) - ro["rgb"]).max(axis=2)
                print(f"  frame {k} {name:13s}: rays {(r['stats'].primaryRays, r['stats'].shadowRays, r['stats'].secondaryRays)} oracle {rays_o}; "
                      f"before bloom: {np.count_nonzero(d > 1e-5)} pixels differ by more than 1e-5 (max {d.max():.3g}), "
                      f"ids differ {np.count_nonzero(r['prim'] != ro['prim'])}")
            a, b = frames["as drawn"]["rgb"], frames["depth-first"]["rgb"]
            print(f"  frame {k} GPU rounds against GPU depth-first: max abs {np.abs(a - b).max():.3g}")
            if bloom:
                thr = c["features"]["bloomFilterThreshold"]
                lum = lambda x: 0.2126 * x[..., 0] + 0.7152 * x[..., 1] + 0.0722 * x[..., 2]
                la, lo = lum(a.astype(np.float64)), lum(ro["rgb"].astype(np.float64))
                near = np.abs(lo - thr) < 1e-5
                print(f"  frame {k} bloom threshold {thr}: {np.count_nonzero((la > thr) != (lo > thr))} pixels on different sides, "
                      f"{np.count_nonzero(near)} pixels of the oracle's frame within 1e-5 of it")


if __name__ == "__main__":
    main()

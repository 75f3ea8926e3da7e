"""profiles/<tag>_counters.md from three rocprofv3 --pmc passes over tools/frame.py (tools/pmc.sh <tag>_l2 / _sq / _grbm).
usage: python tools/counters_md.py <tag> "<what was profiled>" """
import collections, csv, glob, sys

tag, what = sys.argv[1], sys.argv[2]


def load(suffix):
    f = glob.glob(f"gpurun_out/pmc_{tag}_{suffix}/**/p_counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # dispatch -> counter -> value
    name = {}
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = k
    return per, name


sq, sqn = load("sq")
l2, l2n = load("l2")
gr, grn = load("grbm")
# a launch "does work" when it issued more vector instructions than a launch that returns at once (a few per wave)
busy_sq = {d for d, v in sq.items() if v["SQ_INSTS_VALU"] > 40 * max(v["SQ_WAVES"], 1)}


def by_kernel(per, names, only=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for d, v in per.items():
        if only is not None and d not in only:
            continue
        for c, x in v.items():
            agg[names[d]][c] += x
        n[names[d]] += 1
    return agg, n


# dispatch ids are per run: match passes by (kernel, ordinal of the dispatch within that kernel)
def ordinal(names):
    seen, out = collections.Counter(), {}
    for d in sorted(names, key=lambda x: int(x)):
        out[d] = (names[d], seen[names[d]])
        seen[names[d]] += 1
    return out


osq, ol2, ogr = ordinal(sqn), ordinal(l2n), ordinal(grn)
busy_keys = {osq[d] for d in busy_sq}
busy_l2 = {d for d in l2 if ol2[d] in busy_keys}
busy_gr = {d for d in gr if ogr[d] in busy_keys}
A, nA = by_kernel(sq, sqn, busy_sq)
L, nL = by_kernel(l2, l2n, busy_l2)
G, nG = by_kernel(gr, grn, busy_gr)
allN = collections.Counter(sqn.values())
lines = [f"# Counters of the headline frame's kernels ({what})", "",
         "Three separate `rocprofv3 --pmc` passes over `tools/frame.py 16 dragon_glossy_bloom overlapped 1 8` (sixteen frames of the headline configuration",
         "as two launch sequences of eight frames, with the launch sizes of overlapped sequences), via `tools/pmc.sh`: `TCC_HIT_sum TCC_MISS_sum`; `SQ_WAVES SQ_INSTS_VALU",
         "SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY`; `GRBM_GUI_ACTIVE`.  Averages per dispatch over the dispatches",
         "that do work (launches that return at once are counted in the second column only).  L2 hit = HIT / (HIT + MISS); vector issue = 4 x",
         "SQ_ACTIVE_INST_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs); active lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); waves per SIMD =",
         "4 x SQ_WAVE_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8); waiting = SQ_WAIT_ANY / SQ_WAVE_CYCLES.", "",
         "| kernel | dispatches with work / all (2 sequences of 8 frames) | L2 hit | vector issue | active lanes | waves/SIMD | waiting | VALU insts | GUI_ACTIVE |",
         "|---|---|---|---|---|---|---|---|---|"]
for k in sorted(A, key=lambda k: -A[k]["SQ_WAVE_CYCLES"]):
    n = nA[k]
    a = {c: x / n for c, x in A[k].items()}
    hit = L[k]["TCC_HIT_sum"] / max(L[k]["TCC_HIT_sum"] + L[k]["TCC_MISS_sum"], 1) if nL[k] else float("nan")
    gui = G[k]["GRBM_GUI_ACTIVE"] / nG[k] if nG[k] else float("nan")
    simd_cycles = 1024 * gui / 8
    lines.append(f"| `{k}` | {n} / {allN[k]} | {100 * hit:.0f} % | {100 * 4 * a['SQ_ACTIVE_INST_VALU'] / simd_cycles:.0f} % | "
                 f"{100 * a['SQ_THREAD_CYCLES_VALU'] / (64 * a['SQ_ACTIVE_INST_VALU']):.0f} % | {4 * a['SQ_WAVE_CYCLES'] / simd_cycles:.1f} | "
                 f"{100 * a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.0f} % | {a['SQ_INSTS_VALU']:.3g} | {gui:.3g} |")
open(f"profiles/{tag}_counters.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

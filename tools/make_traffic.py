"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh.

HBM bytes of ONE frame's share of a launch sequence of eight frames (all its kernels): rocprofv3 reports both counters in KiB per dispatch;
reads are doubled as prescribed for gfx950 (FETCH_SIZE tallies 128-byte requests at 64 bytes)."""
import collections, csv, json, subprocess, sys
tag = sys.argv[1]
commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
frames = 16  # tools/frame.py 16 dragon_glossy_bloom overlapped 1 8
def total(path):
    t = 0.0
    for r in csv.DictReader(open(path)):
        if "rocclr" in r["Kernel_Name"] or "at::native" in r["Kernel_Name"]:
            continue
        t += float(r["Counter_Value"])
    return t
fetch = total(f"gpurun_out/{tag}/pmc_fetch.csv") * 1024 / frames
write = total(f"gpurun_out/{tag}/pmc_write.csv") * 1024 / frames
out = {"dragon_glossy_bloom": 2 * fetch + write,
       "_measured": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh {tag}, library built from commit {commit} (or its working tree)",
       "_detail": {"FETCH_SIZE_bytes_per_frame_raw": fetch, "read_correction": 2.0, "WRITE_SIZE_bytes_per_frame": write,
                   "source": f"profiles/{tag}_pmc_fetch.csv, profiles/{tag}_pmc_write.csv (separate rocprofv3 --pmc passes over tools/frame.py 16 dragon_glossy_bloom overlapped 1 8: two launch sequences of eight frames)"}}


# Per kernel: time per frame from the timeline of ONE launch sequence of eight frames with the GPU to itself (tools/round_timeline.sh
# <tag>_tl_seq dragon_glossy_bloom alone 1 8: its launches run one after the other, so their durations add up to the sequence), HBM bytes
# per frame from the two counter passes, and the fraction of the HBM roof the two give.
import re
def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
def by_kernel(path):
    t = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        t[short(r["Kernel_Name"])] += float(r["Counter_Value"]) * 1024
    return t
kf, kw = by_kernel(f"gpurun_out/{tag}/pmc_fetch.csv"), by_kernel(f"gpurun_out/{tag}/pmc_write.csv")
dur, calls = collections.defaultdict(float), collections.Counter()
seq_frames = 8
for line in open(f"gpurun_out/{tag}_tl_seq/timeline.txt"):
    m = re.match(r"\s*\d+\s+[\d.]+ \+\s*([\d.]+) gap\s+[\d.-]+ (.*)", line)
    if m:
        dur[m.group(2).strip()] += float(m.group(1))
        calls[m.group(2).strip()] += 1
rows = []
for k, us in sorted(dur.items(), key=lambda kv: -kv[1]):
    if "rocclr" in k or "at::native" in k:
        continue
    ms = us / 1e3 / seq_frames
    bts = (2 * kf.get(k, 0.0) + kw.get(k, 0.0)) / frames
    rows.append({"kernel": k, "ms_per_frame": round(ms, 4), "launches_per_frame": round(calls[k] / seq_frames, 2),
                 "hbm_bytes_per_frame": round(bts), "hbm_frac": round(bts / (ms * 1e-3) / 8e12, 4) if ms > 0 else None})
out["_per_kernel"] = {"source": f"profiles/{tag}_timeline_one_sequence_of_8_alone.txt (one launch sequence of eight frames, nothing else on the GPU) and the two "
                                "counter passes; hbm_frac = bytes / time / 8 TB/s", "rows": rows}
# What bounds the frame, from the counters of the same sixteen frames (profiles/<tag>_counters.md)
lim = {}
try:
    for line in open(f"profiles/{tag}_counters.md"):
        c = [x.strip() for x in line.split("|")]
        if len(c) > 8 and c[1].startswith("`"):
            lim[c[1].strip("`")] = {"l2_hit": c[3], "vector_issue": c[4], "active_lanes": c[5], "waves_per_simd": c[6], "waiting": c[7]}
except OSError:
    pass
total_ms = sum(r["ms_per_frame"] for r in rows)
top = rows[:3]
out["_limiter"] = {"bound": "latency / occupancy of the traversal kernels, not HBM",
                   "why": "the kernels below hold %.0f %% of the frame's kernel time; their waves wait for memory most of their cycles at 1-3 waves per SIMD "
                          "while the frame's HBM traffic is a fraction of the roof (hbm_traffic_frac)" % (100 * sum(r["ms_per_frame"] for r in top) / max(total_ms, 1e-9)),
                   "kernels": [{"kernel": r["kernel"], "share_of_kernel_time": round(r["ms_per_frame"] / max(total_ms, 1e-9), 3),
                                **next((v for k, v in lim.items() if r["kernel"][:40] == k[:40]), {})} for r in top],
                   "source": f"profiles/{tag}_counters.md"}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])

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
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(out)

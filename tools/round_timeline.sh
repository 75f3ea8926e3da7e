#!/bin/bash
# usage (on the GPU box): tools/round_timeline.sh <tag> [frame.py args]  -> the last frame's launches in order: start, duration, gap before, name
set -o pipefail
tag=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o p -- python3 $root/tools/frame.py ${TIMELINE_FRAMES:-3} "$@" > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "camera_kernel" in r["Kernel_Name"]]
fr = rows[idx[-1]:]
t0 = int(fr[0]["Start_Timestamp"]); prev_end = t0
with open(sys.argv[1] + "/timeline.txt", "w") as o:
    for i, r in enumerate(fr):
        n = r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        o.write("%4d %9.1f +%8.1f gap %6.1f %s\n" % (i, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n))
        prev_end = max(prev_end, e)
    o.write("frame %.1f us\n" % ((prev_end - t0) / 1e3))
PY
rm -rf $out/kt
tail -1 $out/timeline.txt

#!/bin/bash
# usage (on the GPU box): tools/kt.sh <tag> [frames] [workload]  -> gpurun_out/<tag>/kernel_stats.csv (+ top rows on stdout)
# per-kernel times of `frames` frames of a bench workload (tools/frame.py) under rocprofv3 --kernel-trace --stats
set -o pipefail
tag=$1; frames=${2:-3}; wl=${3:-dragon_glossy_bloom}; shift; shift; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o p -- python3 $root/tools/frame.py $frames $wl "$@" > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
cp $out/kt/p_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/kt
python3 - $out/kernel_stats.csv $frames <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total {tot/1e6/n:.3f} ms/frame over {n} frames")
for r in rows[:16]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{float(r['TotalDurationNs'])/1e6/n:8.3f} ms/frame  {int(r['Calls'])/n:6.1f} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {name}")
PY

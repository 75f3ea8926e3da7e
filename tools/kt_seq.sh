#!/bin/bash
# usage (on the GPU box): tools/kt_seq.sh <tag> K:S[:o]  -> per-kernel times of tools/frames_probe.py under rocprofv3 --kernel-trace --stats
set -o pipefail
tag=$1; spec=$2
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o p -- python3 $root/tools/frames_probe.py $spec > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
cp $out/kt/p_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/kt
grep "frames per" $out/kt.log
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms")
for r in rows[:14]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{100*float(r['TotalDurationNs'])/tot:6.1f} %  {int(r['Calls']):6d} calls  avg {float(r['AverageNs'])/1e3:9.1f} us  {name}")
PY

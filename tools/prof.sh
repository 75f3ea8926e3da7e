#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]  -> gpurun_out/prof_<tag>/ with rocprofv3 kernel stats
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --cpu-seconds 0 "$@" > $out.log 2>&1
rc=$?
grep -h '"metric"' $out.log | cut -c1-140
python3 - <<PY
import csv,glob
f=glob.glob("$out/**/p_kernel_stats.csv",recursive=True)
for r in csv.DictReader(open(f[0])):
    print("%-60s calls %6s total %10.1f us avg %9.1f us  %5s%%"%(r["Name"][:60],r["Calls"],float(r["TotalDurationNs"])/1e3,float(r["AverageNs"])/1e3,r["Percentage"]))
PY
exit $rc

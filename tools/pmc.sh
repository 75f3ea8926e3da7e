#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [frame.py args] -> per-kernel sums of the counters
tag=$1; ctrs=$2; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/frame.py ${PMC_FRAMES:-2} "$@" > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
python3 - <<PY
import csv,glob,collections
f=glob.glob("$out/**/p_counter_collection.csv",recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
seen=set()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"].replace("void (anonymous namespace)::","").replace("(anonymous namespace)::","").split("(")[0][:40]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); calls[k]+=1
for k,v in agg.items():
    print("%-42s n=%4d "%(k,calls[k])+"  ".join("%s=%.4g"%(c,x/calls[k]) for c,x in v.items()))
PY

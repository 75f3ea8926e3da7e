#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/lat_$1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/tools/latency_probe.py > $out.log 2>&1 || { tail $out.log; exit 1; }
grep probe $out.log
python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$out/p_kernel_trace.csv"))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows:
    n=r["Kernel_Name"]
    if "trace_kernel" in n:
        print("%-45s grid %8s  %9.1f us"%(n.replace("void (anonymous namespace)::","").split("(")[0][:45], r["Grid_Size"] if "Grid_Size" in r else "", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY

#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc_fetch.csv,pmc_write.csv}
set -o pipefail
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout -k 10 400 python3 bench.py > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cut -c1-200 $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o p -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-seconds 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
cp $out/kt/p_kernel_stats.csv $out/kernel_stats.csv
grep -h '"metric"' $out/kt.log > $out/bench_under_rocprof.json
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o p -- python3 $root/tools/frame.py 2 > $out/pf.log 2>&1 || { tail -5 $out/pf.log; exit 1; }
cp $out/pf/p_counter_collection.csv $out/pmc_fetch.csv
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o p -- python3 $root/tools/frame.py 2 > $out/pw.log 2>&1 || { tail -5 $out/pw.log; exit 1; }
cp $out/pw/p_counter_collection.csv $out/pmc_write.csv
rm -rf $out/kt/p_kernel_trace.csv
echo done

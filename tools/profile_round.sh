#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh <tag>
#   -> gpurun_out/<tag>/{bench.json, kernel_stats.csv (one frame at a time), kernel_stats_sequence.csv (launch sequences of 8 frames, one
#      at a time), kernel_stats_overlapped.csv (bench defaults), bench_under_rocprof*.json, pmc_fetch.csv, pmc_write.csv (16 frames in two
#      launch sequences of 8)}
set -o pipefail
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/$tag
mkdir -p $out
cd $root
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
cut -c1-200 $out/bench.json
cd /tmp && export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
# per-kernel durations of frames rendered one at a time: they add up to the frame's time alone
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o p -- python3 $root/bench.py --frames-in-flight 1 --frames-per-sequence 1 --steps 5 --warmup 1 --cpu-seconds 0 > $out/kt.log 2>&1 || { tail -5 $out/kt.log; exit 1; }
cp $out/kt/p_kernel_stats.csv $out/kernel_stats.csv
grep -h '"metric"' $out/kt.log > $out/bench_under_rocprof.json
# launch sequences of 8 frames, one sequence at a time
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kts -o p -- python3 $root/bench.py --frames-in-flight 1 --frames-per-sequence 8 --steps 16 --warmup 8 --cpu-seconds 0 > $out/kts.log 2>&1 || { tail -5 $out/kts.log; exit 1; }
cp $out/kts/p_kernel_stats.csv $out/kernel_stats_sequence.csv
grep -h '"metric"' $out/kts.log > $out/bench_under_rocprof_sequence.json
# the bench's defaults: sequences on three streams (four from 32 steps on), kernels of different sequences overlap, durations stretch
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kto -o p -- python3 $root/bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $out/kto.log 2>&1 || { tail -5 $out/kto.log; exit 1; }
cp $out/kto/p_kernel_stats.csv $out/kernel_stats_overlapped.csv
grep -h '"metric"' $out/kto.log > $out/bench_under_rocprof_overlapped.json
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pf -o p -- python3 $root/tools/frame.py 16 dragon_glossy_bloom overlapped 1 8 > $out/pf.log 2>&1 || { tail -5 $out/pf.log; exit 1; }
cp $out/pf/p_counter_collection.csv $out/pmc_fetch.csv
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pw -o p -- python3 $root/tools/frame.py 16 dragon_glossy_bloom overlapped 1 8 > $out/pw.log 2>&1 || { tail -5 $out/pw.log; exit 1; }
cp $out/pw/p_counter_collection.csv $out/pmc_write.csv
rm -rf $out/kt/p_kernel_trace.csv $out/kto/p_kernel_trace.csv $out/kts/p_kernel_trace.csv
echo done

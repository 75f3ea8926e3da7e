#!/bin/bash
# usage (on the GPU box): tools/bench_lines.sh <tag>  -> gpurun_out/<tag>/bench_<workload>.json for the non-headline workloads
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
for w in dragon_primary_shadow teapot_monkey_phong cornell_mirror dragon_4k_orbit dragon_glossy1_bloom dragon_glossy16_bloom; do
  extra=""
  if [ $w = dragon_glossy16_bloom ]; then extra="--frames-in-flight 2 --steps 6 --warmup 2"; fi  # 640 M rays and 40 GB of task queues per frame
  timeout -k 10 900 python3 bench.py --workload $w --cpu-seconds 0 $extra > $out/bench_$w.json 2> $out/bench_$w.err || { tail -5 $out/bench_$w.err; exit 1; }
  cut -c1-200 $out/bench_$w.json
done

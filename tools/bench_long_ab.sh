#!/bin/bash
# usage (on the GPU box): tools/bench_long_ab.sh  -> 96 steps with and without the staggered first wave, streams of unequal priority
set -o pipefail
for rep in 1 2; do
  for a in "" "--sequence-sizes 8,8,8,8,8,8,8,8,8,8,8,8" "--stream-priorities=0"; do
    echo -n "[$a] "
    timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 96 --warmup 8 --cpu-seconds 0 $a 2>&1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per step, %.0f Mrays/s' % (d['ms_per_step'], d['value']))" || exit 1
  done
done

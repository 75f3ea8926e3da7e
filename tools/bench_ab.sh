#!/bin/bash
# usage (on the GPU box): tools/bench_ab.sh "<args A>" "<args B>" [reps]  -> bench.py at the driver's 20 steps / 5 warm-up with either argument set, alternating
set -o pipefail
for rep in $(seq 1 ${3:-2}); do
  for a in "$1" "$2"; do
    echo -n "[$a] "
    timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 $a 2>&1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per step, %.0f Mrays/s' % (d['ms_per_step'], d['value']))" || exit 1
  done
done

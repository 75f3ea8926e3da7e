#!/bin/bash
# usage (on the GPU box): tools/bench_sizes.sh <sizes>...  -> bench.py at 20 steps / 5 warm-up with each list of sequence sizes ("" = default), twice, alternating
set -o pipefail
for rep in 1 2; do
  for a in "$@"; do
    echo -n "[$a] "
    timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 ${a:+--sequence-sizes $a} 2>&1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms per step, %.0f Mrays/s' % (d['ms_per_step'], d['value']))" || exit 1
  done
done

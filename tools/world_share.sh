#!/bin/bash
# usage (on the GPU box): tools/world_share.sh  -> one rank's share of 20 frames as two sequences of ten (render + bloom, no copy) for worlds of 2, 4, 8,
# the two streams at unequal priorities as bench.py creates them
set -o pipefail
export RT_PROBE_NO_COPY=1 RT_PROBE_PRIORITIES=-1,0
for w in 2 4 8; do
  echo -n "world $w: "
  RT_PROBE_WORLD=$w timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/frames_probe.py 10:2:o 2>&1 | grep "frames per" || exit 1
done

#!/bin/bash
# usage (on the GPU box): tools/world_prio.sh  -> one rank's share of 20 frames (two sequences of ten) for worlds of 2, 4, 8, streams equal and unequal
set -o pipefail
for w in 2 4 8; do
  for p in none -1,0; do
    if [ "$p" = none ]; then unset RT_PROBE_PRIORITIES; else export RT_PROBE_PRIORITIES=$p; fi
    echo -n "world $w priorities $p: "
    RT_PROBE_WORLD=$w timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/frames_probe.py 10:2:o 2>&1 | grep "frames per" || exit 1
  done
done

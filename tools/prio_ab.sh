#!/bin/bash
# usage (on the GPU box): tools/prio_ab.sh <spec> <priorities>...  -> the frames probe with default streams and with each priority list, twice, alternating
set -o pipefail
spec=$1; shift
for rep in 1 2; do
  for p in none "$@"; do
    if [ "$p" = none ]; then unset RT_PROBE_PRIORITIES; else export RT_PROBE_PRIORITIES=$p; fi
    echo -n "priorities $p: "
    timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/frames_probe.py $spec 2>&1 | grep "frames per" || exit 1
  done
done

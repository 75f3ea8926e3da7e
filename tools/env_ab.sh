#!/bin/bash
# usage (on the GPU box): tools/env_ab.sh <probe spec> <VAR=value>...  -> tools/frames_probe.py without and with each setting, twice, alternating
set -o pipefail
spec=$1; shift
for rep in 1 2; do
  for e in none "$@"; do
    echo -n "$e: "
    if [ "$e" = none ]; then timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/frames_probe.py $spec 2>&1 | grep "frames per" | tr '\n' ' ' || exit 1
    else env $e timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/frames_probe.py $spec 2>&1 | grep "frames per" | tr '\n' ' ' || exit 1; fi
    echo
  done
done

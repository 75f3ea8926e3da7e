#!/bin/bash
# usage (on the GPU box): tools/lib_ab.sh <probe spec> <lib>...  -> tools/frames_probe.py with the in-tree library ("tree") and with each variant library, twice, alternating
set -o pipefail
spec=$1; shift
for rep in 1 2; do
  for l in tree "$@"; do
    if [ "$l" = tree ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$GRAFT_REPO_ROOT/variants_lib/$l; fi
    echo -n "$l: "
    timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/frames_probe.py $spec 2>&1 | grep "frames per" | tr '\n' ' ' || exit 1
    echo
  done
done

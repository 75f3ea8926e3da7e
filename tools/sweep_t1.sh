#!/bin/bash
# sweep of the lane/group switch point and query budget on the headline frame
cd /root/repo
for t1 in 65536 131072 196608 262144 393216; do
  for qb in 24; do
    echo "T1=$t1 QB=$qb"
    RT_AMD_T1=$t1 RT_AMD_QBUDGET=$qb timeout -k 10 120 python bench.py --steps 20 --warmup 3 --cpu-seconds 0 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" || exit 1
  done
done

#!/bin/bash
# usage (on the GPU box): tools/sweep_env.sh "<bench args>" "VAR=a VAR2=b" "VAR=c" ...   -> one line per environment
args=$1; shift
for cfg in "$@"; do
  r=$(env $cfg timeout -k 10 150 python3 bench.py $args --cpu-seconds 0 2>/dev/null | grep -o "\"value\": [0-9.]*\|\"ms_per_step\": [0-9.]*" | tr "\n" " ")
  echo "$cfg => $r"
done

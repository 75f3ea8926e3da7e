# usage (on the GPU box): tools/seq_sizes.sh <steps> <sizes> ...   bench value for each split of the steps into launch sequences
cd $GRAFT_REPO_ROOT
steps=$1; shift
for sz in "$@"; do
  timeout -k 10 300 python bench.py --steps $steps --warmup 8 --cpu-seconds 0 --sequence-sizes $sz > gpurun_out/tmp_b.json 2>gpurun_out/tmp_b.err || { tail -3 gpurun_out/tmp_b.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/tmp_b.json')); print('$sz', 'value', round(d['value']), 'ms', round(d['ms_per_step'],3))"
done

cd $GRAFT_REPO_ROOT
for sz in "5,5,5,5" "6,5,4,3,2" "7,6,4,2,1" "8,6,4,2" "4,4,4,4,2,2" "6,6,4,2,1,1" "8,7,5" "8,8,4" "3,3,3,3,2,2,2,2"; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --sequence-sizes $sz > gpurun_out/tmp_b.json 2>gpurun_out/tmp_b.err || { tail -3 gpurun_out/tmp_b.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/tmp_b.json')); print('$sz', 'value', round(d['value']), 'ms', round(d['ms_per_step'],3))"
done

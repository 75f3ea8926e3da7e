# usage (on the GPU box): tools/ks_scan.sh <workload> "K S" ...  -> bench value for each split of 24 steps
cd $GRAFT_REPO_ROOT
wl=$1; shift
for kf in "$@"; do set -- $kf
  timeout -k 10 300 python bench.py --workload $wl --steps 24 --warmup 4 --cpu-seconds 0 --frames-per-sequence $1 --frames-in-flight $2 > gpurun_out/tmp_b.json 2>gpurun_out/tmp_b.err || { tail -3 gpurun_out/tmp_b.err; exit 1; }
  python -c "import json; d=json.load(open('gpurun_out/tmp_b.json')); print('$wl K=$1 S=$2', 'value', round(d['value']), 'ms', round(d['ms_per_step'],3))"
done

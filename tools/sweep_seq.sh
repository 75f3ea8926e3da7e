cd $GRAFT_REPO_ROOT
spec=${SPEC:-8:4:o}
run() { echo "== $*"; env "$@" timeout -k 10 200 python tools/frames_probe.py $spec 2>&1 | grep "frames per" || exit 1; }
while read -r line; do [ -n "$line" ] && run $line; done

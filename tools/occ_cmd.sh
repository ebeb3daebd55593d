set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/hnsw-slim_amd
run() { HS_LIB=$1 python bench.py --index-dir /tmp/idx --ef $3 --streams 4 --steps 20 --warmup 4 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('RESULT $1 ef=$3 $2', d['value'], d['ms_per_step'], d['roofline']['launch_ms'])"; }
for ef in 72 96; do
run $R/libhnsw_slim_amd.so "" $ef
run $R/libhnsw_slim_amd_w5.so "" $ef
run $R/libhnsw_slim_amd_w6.so "" $ef
done

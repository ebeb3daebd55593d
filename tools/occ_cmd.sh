set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/hnsw-slim_amd
run() { HS_LIB=$1 python bench.py --index-dir /tmp/idx --ef 96 --streams 4 --steps 20 --warmup 4 --no-cpu-baseline $2 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('RESULT $1 $2', d['value'], d['ms_per_step'], d['roofline']['single_launch_ms'])"; }
run $R/libhnsw_slim_amd.so "--cand-cap 448"
run $R/libhnsw_slim_amd_w4.so "--cand-cap 448"
run $R/libhnsw_slim_amd_w5.so "--cand-cap 448"
run $R/libhnsw_slim_amd_w4.so "--cand-cap 448 --hash-slots 1152"
run $R/libhnsw_slim_amd_w5.so "--cand-cap 448 --hash-slots 1152"

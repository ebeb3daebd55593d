# Round-end refresh (lighter than prof_cmd.sh): the bench line as the driver runs it, the strictly serial run, and the
# rocprofv3 kernel-trace stats of the serial run (whose average fast_kernel duration must agree with roofline.launch_ms).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
python bench.py --index-dir /tmp/idx > $O/bench_pipelined.json 2> $O/bench_pipelined.log || exit 1
EF=$(python -c "import json;print(json.load(open('$O/bench_pipelined.json'))['config']['ef_search'])")
python bench.py --index-dir /tmp/idx --ef $EF --streams 1 --no-cpu-baseline > $O/bench_1stream.json 2> $O/bench_1stream.log || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_1stream -- python bench.py --index-dir /tmp/idx --ef $EF --streams 1 --no-cpu-baseline > /dev/null 2>&1
find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -delete
echo EF=$EF

# Round-end measurement recipe (run through gpurun): the bench line, a strictly serial run, rocprofv3
# kernel-trace stats of both, and the PMC passes for HBM traffic (separate --pmc runs, as the guide asks).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r01
mkdir -p $O
python bench.py --index-dir /tmp/idx --steps 20 --warmup 4 > $O/bench_pipelined.json 2> $O/bench_pipelined.log
EF=$(python -c "import json;print(json.load(open('$O/bench_pipelined.json'))['config']['ef_search'])")
python bench.py --index-dir /tmp/idx --ef $EF --streams 1 --steps 20 --warmup 4 --no-cpu-baseline > $O/bench_1stream.json 2> $O/bench_1stream.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pipelined -- python bench.py --index-dir /tmp/idx --ef $EF --steps 20 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_1stream -- python bench.py --index-dir /tmp/idx --ef $EF --streams 1 --steps 20 --warmup 4 --no-cpu-baseline > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=$O/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --output-format csv -d $D -- python bench.py --index-dir /tmp/idx --ef $EF --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_calib.hip -o /tmp/gather_calib
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib -- /tmp/gather_calib A > $O/gather_calib.log 2>&1
find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -size +20M -delete
echo EF=$EF

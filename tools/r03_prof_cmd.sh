#!/bin/bash
# Round-3 measurement recipe (run through gpurun): the bench line as the driver runs it, the strictly serial variant, rocprofv3
# kernel-trace stats of both, PMC passes for HBM traffic and the instruction mix (separate --pmc runs, as the guide asks) and
# the gather calibration of FETCH_SIZE.  tools/summarize_prof.py r03 turns the raw output into profiles/r03_*.
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
export TMPDIR=/tmp
O=gpurun_out/r03
rm -rf $O; mkdir -p $O
I=/tmp/hsidx
python bench.py --index-dir $I --steps 20 --warmup 5 > $O/bench_pipelined.json 2> $O/bench_pipelined.log || { tail -5 $O/bench_pipelined.log; kill $HB; exit 1; }
EF=$(python -c "import json;print(json.load(open('$O/bench_pipelined.json'))['config']['ef_search'])")
python bench.py --index-dir $I --ef $EF --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_1stream.json 2> $O/bench_1stream.log || { kill $HB; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_pipelined -- python bench.py --index-dir $I --ef $EF --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_1stream -- python bench.py --index-dir $I --ef $EF --streams 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  D=$O/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --output-format csv -d $D -- python bench.py --index-dir $I --ef $EF --streams 1 --steps 1 --warmup 1 --batches-per-step 20 --no-cpu-baseline > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_insts -- python bench.py --index-dir $I --ef $EF --streams 1 --steps 1 --warmup 1 --batches-per-step 20 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/pmc_wait -- python bench.py --index-dir $I --ef $EF --streams 1 --steps 1 --warmup 1 --batches-per-step 20 --no-cpu-baseline > /dev/null 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_calib.hip -o /tmp/gather_calib
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_calib -- /tmp/gather_calib A > $O/gather_calib.log 2>&1
find $O -name "*agent_info.csv" -delete
find $O -name "*kernel_trace.csv" -delete
kill $HB
echo EF=$EF
python tools/summarize_prof.py r03 2>&1 | tail -40

set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --index-dir /tmp/idx --steps 10 --warmup 2 > gpurun_out/bench_r1k.json 2> gpurun_out/bench_r1k.log
grep "ef=" gpurun_out/bench_r1k.log
python -c "
import json;d=json.load(open('gpurun_out/bench_r1k.json'));print(d['value'],d['ms_per_step'],d['config']['ef_search'],d['config']['recall_at_10'],d['roofline'],d['cpu_baseline'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python bench.py --index-dir /tmp/idx --ef 96 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_kt.json 2> gpurun_out/prof_kt.log
find gpurun_out/prof_kt -name "*kernel_stats*" | head -2
cat $(find gpurun_out/prof_kt -name "*kernel_stats.csv" | head -1) | head -8
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof_pmc1 -- python bench.py --index-dir /tmp/idx --ef 96 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc1.json 2> gpurun_out/prof_pmc1.log
ls gpurun_out/prof_pmc1/*/ | head

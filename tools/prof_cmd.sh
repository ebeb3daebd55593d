# Round-end measurement recipe (run through gpurun): bench at S=4 and S=1, then rocprofv3 kernel trace.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --index-dir /tmp/idx --steps 20 --warmup 4 > gpurun_out/bench_s4.json 2> gpurun_out/bench_s4.log
grep "ef=" gpurun_out/bench_s4.log
python -c "
import json;d=json.load(open('gpurun_out/bench_s4.json'));print(d['value'],d['ms_per_step'],d['config']['ef_search'],d['config']['recall_at_10'],d['roofline'],d['cpu_baseline'])"
python bench.py --index-dir /tmp/idx --ef 96 --streams 1 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/bench_s1.json 2> gpurun_out/bench_s1.log
python -c "
import json;d=json.load(open('gpurun_out/bench_s1.json'));print(d['value'],d['ms_per_step'],d['roofline'])"
python bench.py --index-dir /tmp/idx --ef 96 --streams 8 --steps 24 --warmup 8 --no-cpu-baseline > gpurun_out/bench_s8.json 2> gpurun_out/bench_s8.log
python -c "
import json;d=json.load(open('gpurun_out/bench_s8.json'));print(d['value'],d['ms_per_step'],d['roofline'])"
python bench.py --index-dir /tmp/idx --ef 64 --streams 4 --steps 20 --warmup 4 --no-cpu-baseline > gpurun_out/bench_s4_ef64.json 2> gpurun_out/bench_s4_ef64.log
python -c "
import json;d=json.load(open('gpurun_out/bench_s4_ef64.json'));print(d['value'],d['ms_per_step'],d['roofline'])"

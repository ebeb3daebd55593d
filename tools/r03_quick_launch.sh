#!/bin/bash
# round 3: quick check of one change: parity of the kernel variants + the single-stream launch time and the pipelined value
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -x -q > gpurun_out/r03_quick_tests.log 2>&1; tail -2 gpurun_out/r03_quick_tests.log
python bench.py --index-dir /tmp/hsidx --ef 70 --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_quick_1stream.json 2> gpurun_out/r03_quick.log || tail -5 gpurun_out/r03_quick.log
python bench.py --index-dir /tmp/hsidx --ef 70 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_quick_pipelined.json 2>> gpurun_out/r03_quick.log || tail -5 gpurun_out/r03_quick.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03_quick_kt -- python bench.py --index-dir /tmp/hsidx --ef 70 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import json, glob
for f in ("r03_quick_1stream", "r03_quick_pipelined"):
    j = json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
    print(f, "value", j["value"], "frac", j["roofline"]["frac"], "launch_ms", j["roofline"]["launch_ms"], "dev-resident", j["config"].get("device_resident_pipelined_qps"))
for f in glob.glob("gpurun_out/r03_quick_kt/**/*kernel_stats.csv", recursive=True):
    for l in open(f):
        if "hs::" in l and "bf_" not in l: print(l.strip()[:200])
PY
find gpurun_out/r03_quick_kt -name "*agent_info.csv" -delete; find gpurun_out/r03_quick_kt -name "*kernel_trace.csv" -delete

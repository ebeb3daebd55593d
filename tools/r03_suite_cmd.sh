#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x --durations=5 > gpurun_out/r03_t4.log 2>&1; rc=$?; tail -14 gpurun_out/r03_t4.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --index-dir /tmp/hsidx --steps 20 --warmup 5 --no-cpu-baseline --ef 68 > gpurun_out/r03_b4.json 2> gpurun_out/r03_b4.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_b4.json"))
print("value %.2fM launch_ms %.3f frac %.3f devres %.2fM" % (d["value"]/1e6, d["roofline"]["launch_ms"], d["roofline"]["frac"], d["config"]["device_resident_pipelined_qps"]/1e6))
PY
exit $rc

#!/bin/bash
mkdir -p gpurun_out
run() {  # name envs...
  n=$1; shift
  env "$@" timeout -k 10 300 python bench.py --index-dir /tmp/hsidx --steps 10 --warmup 3 --no-cpu-baseline --ef 68 > gpurun_out/r03_ord_$n.json 2> gpurun_out/r03_ord_$n.log || { echo "$n failed"; tail -3 gpurun_out/r03_ord_$n.log; return; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03_ord_$n.json"))
print("$n: value %.2f M q/s, launch_ms %.3f, frac %.3f, dev-resident pipelined %.2f M" % (d["value"]/1e6, d["roofline"]["launch_ms"], d["roofline"]["frac"], d["config"]["device_resident_pipelined_qps"]/1e6))
PY
}
run ordered HS_X=1
run unordered HS_ORDER=0
run split_only HS_ORDER=2

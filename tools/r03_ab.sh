#!/bin/bash
# A/B of flat-kernel build variants on one index: single launch + pipelined rate at the operating point and at ef=128
mkdir -p gpurun_out
R=$PWD/hnsw-slim_amd
run() {  # name lib waves_per_cu ef
  HS_LIB=$2 HS_FLAT_WAVES_PER_CU=$3 timeout -k 10 300 python bench.py --index-dir /tmp/hsidx --steps 10 --warmup 3 --no-cpu-baseline --ef $4 > gpurun_out/r03_ab_$1_$4.json 2> gpurun_out/r03_ab_$1_$4.log || { echo "$1 failed"; tail -3 gpurun_out/r03_ab_$1_$4.log; return; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r03_ab_$1_$4.json"))
print("$1 ef=$4: value %.2f M q/s, launch_ms %.3f, frac %.3f, dev-resident pipelined %.2f M, kernel %s" % (d["value"]/1e6, d["roofline"]["launch_ms"], d["roofline"]["frac"], d["config"]["device_resident_pipelined_qps"]/1e6, d["roofline"]["kernel"]))
PY
}
for ef in 68 128; do
  run main $R/libhnsw_slim_amd.so 20 $ef
  run w5i $R/libhnsw_slim_amd_w5i.so 20 $ef
  run w4 $R/libhnsw_slim_amd_w4.so 16 $ef
  run w6 $R/libhnsw_slim_amd_w6.so 24 $ef
done

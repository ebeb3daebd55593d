#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r03_t2.log 2>&1
rc=$?
tail -3 gpurun_out/r03_t2.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
timeout -k 10 500 python bench.py --index-dir /tmp/hsidx --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_b2_flat.json 2> gpurun_out/r03_b2_flat.log
rc2=$?
grep "ef=" gpurun_out/r03_b2_flat.log | cut -c1-200
if [ $rc2 -ge 124 ]; then echo "bench killed rc=$rc2"; exit $rc2; fi
timeout -k 10 300 python tools/flat_diag.py /tmp/hsidx 32,48,68,96,128,256 > gpurun_out/r03_diag2.log 2>&1
cat gpurun_out/r03_diag2.log
exit $rc

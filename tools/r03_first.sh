#!/bin/bash
# round 3, first GPU call: whole parity suite with the flat kernel as the default, then bench A/B flat vs lean on one index
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/r03_t1.log 2>&1
rc=$?
tail -5 gpurun_out/r03_t1.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
timeout -k 10 500 python bench.py --index-dir /tmp/hsidx --steps 20 --warmup 5 > gpurun_out/r03_b_flat.json 2> gpurun_out/r03_b_flat.log
rc2=$?
tail -3 gpurun_out/r03_b_flat.log
if [ $rc2 -ge 124 ]; then echo "bench killed rc=$rc2"; exit $rc2; fi
HS_KERNEL=lean timeout -k 10 400 python bench.py --index-dir /tmp/hsidx --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03_b_lean.json 2> gpurun_out/r03_b_lean.log
tail -3 gpurun_out/r03_b_lean.log
exit $rc

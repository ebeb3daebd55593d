#!/bin/bash
# round 3: d = 960 with a launch that FILLS the chip (10k queries instead of configs[2]'s 1k): flat kernel (2 waves per SIMD at
# d = 960) against the fast kernel (3 waves per SIMD), same index (GIST-like, 200k rows)
export TMPDIR=/tmp IDX_DIR=/tmp/gist_big NQ=10000
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_gist_bigbatch.log
echo "== flat (default)" > $O
EFS=64,256,384 python -u tools/other_configs.py gist 200000 2>&1 | grep "^ef=\|^build" >> $O
echo "== HS_KERNEL=fast" >> $O
HS_KERNEL=fast EFS=64,256,384 python -u tools/other_configs.py gist 200000 2>&1 | grep "^ef=" >> $O
kill $HB
cut -c1-220 $O

#!/bin/bash
# round 3 A/B (historical: the HS_FLAT_GROW knob existed for this run only and the growth it switched was removed afterwards): should a launch smaller
# than the wave slots take a larger LDS share per query (capi.cpp plan_flat)?
# (a) GIST-like d=960, 1k-query launches: single launch and 16 in flight; (b) SIFT-like bench index, 1250-query launches (flat_diag.py)
export TMPDIR=/tmp
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_grow_ab.log
export IDX_DIR=/tmp/gist_grow
echo "== gist 200k, grow (default)" > $O
EFS=384 python -u tools/other_configs.py gist 200000 2>&1 | grep "^ef=\|^PIPE\|^build" >> $O
echo "== gist 200k, HS_FLAT_GROW=0" >> $O
HS_FLAT_GROW=0 EFS=384 python -u tools/other_configs.py gist 200000 2>&1 | grep "^ef=\|^PIPE" >> $O
unset IDX_DIR
for g in 1 0; do
  echo "== sift 1M, 1250-query launches, HS_FLAT_GROW=$g" >> $O
  HS_FLAT_GROW=$g NQ=1250 python -u tools/flat_diag.py /tmp/hsidx 70 2>&1 | grep "^ef=" | cut -c1-260 >> $O
  HS_FLAT_GROW=$g NQ=625 python -u tools/flat_diag.py /tmp/hsidx 70 2>&1 | grep "^ef=" | cut -c1-260 >> $O
done
kill $HB
cat $O | cut -c1-260

#!/bin/bash
# round 3 A/B: HNSW-SlimQ kernel with / without the next-tile + raw-row touch loads (make slimqvar SLIMQVAR=nopf SLIMQFLAGS=-DHS_SLIMQ_PREFETCH=0)
# on the SIFT-1M-like d=128 index (hnswlib graph) and, second argument "rq", the reference-style graph.   usage: r03_slimq_ab.sh sift|cohere [rq]
W=$1
export GRAPH=${2:-hnswlib} TMPDIR=/tmp
export IDX_DIR=/tmp/slimq_${W}_$GRAPH
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_slimq_ab_${W}_$GRAPH.log
if [ "$W" = "sift" ]; then export EFS=64,128,256,384; else export EFS=64,256,1024; fi
echo "== default (touch loads on)" > $O
python -u tools/slimq_config.py $W >> $O 2>&1 || { tail -5 $O; kill $HB; exit 1; }
echo "== nopf" >> $O
HS_LIB=$PWD/hnsw-slim_amd/libhnsw_slim_amd_nopf.so python -u tools/slimq_config.py $W 2>&1 | grep "^ef=" >> $O
echo "== default again" >> $O
python -u tools/slimq_config.py $W 2>&1 | grep "^ef=" >> $O
kill $HB
grep "^==\|^ef=\|^build\|^graph" $O | cut -c1-300

#!/bin/bash
# round 3: phase stamps of the HNSW-SlimQ kernel (make slimqdiag) on the SIFT-1M-like and, optionally, COHERE-like index   usage: r03_slimq_diag.sh sift|cohere [rq]
W=$1
export GRAPH=${2:-hnswlib} TMPDIR=/tmp
export IDX_DIR=/tmp/slimq_${W}_$GRAPH
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_slimq_diag_${W}_$GRAPH.log
if [ "$W" = "sift" ]; then export EFS=256; D=64,256,384; else export EFS=256; D=64,256,1024; fi
python -u tools/slimq_config.py $W > $O 2>&1 || { tail -5 $O; kill $HB; exit 1; }
DIAG_EF=$D python -u tools/slimq_config.py $W 2>&1 | grep -A1 "^DIAG" >> $O
kill $HB
grep "^ef=\|^build\|^graph\|DIAG\|shader" $O | cut -c1-330

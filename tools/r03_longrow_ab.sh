#!/bin/bash
# round 3 A/B: the flat kernel's long-row distance pass at d = 960 (GIST-like, 1000-query launches): loads in flight per round x
# wavefronts per SIMD (make flatvar FLATVAR=bXXwY FLATFLAGS="-DHS_FLAT_LONG_B=XX -DHS_FLAT_LONG_WAVES=Y"), plus the phase stamps.
N=${1:-200000}
export TMPDIR=/tmp IDX_DIR=/tmp/gist_ab
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_longrow_ab.log
EFS=384 python -u tools/other_configs.py gist $N > $O 2>&1 || { tail -5 $O; kill $HB; exit 1; }
for v in b30w3 b30w2 b30w1 b60w1; do
  echo "== $v" >> $O
  HS_LIB=$PWD/hnsw-slim_amd/libhnsw_slim_amd_$v.so EFS=64,384,512 python -u tools/other_configs.py gist $N 2>&1 | grep "^ef=" >> $O
done
echo "== default + phase stamps" >> $O
DIAG_EF=64,384 python -u tools/other_configs.py gist $N 2>&1 | grep -A1 "^DIAG" >> $O
kill $HB
grep -v "amdgpu.ids" $O | cut -c1-330

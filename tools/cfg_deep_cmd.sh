cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
echo "=== DEEP-10M-like GEN=${GEN:-32768,12,40,4}" > gpurun_out/r2_cfg_deep_a.log
GEN=${GEN:-32768,12,40,4} EFS=32,64,128,256,384,512 python -u tools/other_configs.py deep 10000000 >> gpurun_out/r2_cfg_deep_a.log 2>&1
kill $HB

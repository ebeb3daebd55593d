cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
python -u tools/slimq_config.py cohere > gpurun_out/r2_cfg_cohere_a.log 2>&1
kill $HB

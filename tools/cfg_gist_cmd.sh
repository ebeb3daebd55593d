cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2 > gpurun_out/r2_cfg_gist_c.log
EFS=64,128,256,384 python -u tools/other_configs.py gist >> gpurun_out/r2_cfg_gist_c.log 2>&1
kill $HB

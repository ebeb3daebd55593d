cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_patch.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
echo "=== DEEP-10M-like GEN=32768,12,40,4"
GEN=32768,12,40,4 EFS=32,64,128,256,384,512 python tools/other_configs.py deep 10000000 2>&1 | grep -v "amdgpu.ids"

# development A/B: this build against hnsw-slim_amd/libhs_base.so across batch sizes (one launch each)
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 || exit 1
python tools/qbench.py --efs 70 > /dev/null 2>&1
for NQ in 1250 10000 65536; do
  echo "nq=$NQ base"; HS_LIB=$PWD/hnsw-slim_amd/libhs_base.so python tools/qbench.py --efs 32,70,128 --nq $NQ --reps 5 2>&1 | grep -E "^ef="
  echo "nq=$NQ new";  python tools/qbench.py --efs 32,70,128 --nq $NQ --reps 5 --check 2>&1 | grep -E "^ef=|equal False"
done

cd $GRAFT_REPO_ROOT
L=$PWD/hnsw-slim_amd
python tools/qbench.py --efs 70 > /dev/null 2>&1
echo "fast (default)"; python tools/qbench.py --efs 70 --check 2>&1 | grep -E "^ef=|oracle"; python tools/qbench.py --efs 70 --nq 32768 --reps 5 2>&1 | grep "^ef="
for w in 4 5; do
  echo "lean w$w"; HS_LEAN=1 HS_LIB=$L/libhnsw_slim_amd_l$w.so python tools/qbench.py --efs 70 --check 2>&1 | grep -E "^ef=|oracle"; HS_LEAN=1 HS_LIB=$L/libhnsw_slim_amd_l$w.so python tools/qbench.py --efs 70 --nq 32768 --reps 5 2>&1 | grep "^ef="
done

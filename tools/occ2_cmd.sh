# development aid: steady-state rate (one 32k-query launch) of the fast kernel at 4 / 5 / 6 waves per SIMD builds
cd $GRAFT_REPO_ROOT
L=$PWD/hnsw-slim_amd
python tools/qbench.py --efs 68 > /dev/null 2>&1
echo "w4 default nq=32768"; python tools/qbench.py --efs 68 --nq 32768 --reps 5 2>&1 | grep "^ef="
echo "w4 cand300 hash1024 nq=32768"; python tools/qbench.py --efs 68 --nq 32768 --reps 5 --cand-cap 300 --hash-slots 1024 2>&1 | grep "^ef="
echo "w5 cand300 hash1024 nq=32768"; HS_LIB=$L/libhnsw_slim_amd_w5.so python tools/qbench.py --efs 68 --nq 32768 --reps 5 --cand-cap 300 --hash-slots 1024 2>&1 | grep "^ef="
echo "w6 cand260 hash768 nq=32768"; HS_LIB=$L/libhnsw_slim_amd_w6.so python tools/qbench.py --efs 68 --nq 32768 --reps 5 --cand-cap 260 --hash-slots 768 2>&1 | grep "^ef="
echo "w6 cand300 hash1024 nq=32768"; HS_LIB=$L/libhnsw_slim_amd_w6.so python tools/qbench.py --efs 68 --nq 32768 --reps 5 --cand-cap 300 --hash-slots 1024 2>&1 | grep "^ef="
echo "group nq=32768"; HS_GROUP=1 python tools/qbench.py --efs 68 --nq 32768 --reps 5 2>&1 | grep "^ef="

# development aid: parity on the tie-heavy tests, then the 10k single launch and the steady state of the current build,
# with and without the two-launch ordered fast pass (HS_ORDER)
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5 || exit 1
python tools/qbench.py --efs 70 > /dev/null 2>&1
echo "index order"; HS_ORDER=0 python tools/qbench.py --efs 32,70,128 2>&1 | grep -E "^ef=|oracle"
echo "entry-distance order"; python tools/qbench.py --efs 32,70,128 --check 2>&1 | grep -E "^ef=|oracle"
echo "steady, index order"; HS_ORDER=0 python tools/qbench.py --efs 70 --nq 65536 --reps 3 2>&1 | grep -E "^ef="
echo "steady, entry-distance order"; python tools/qbench.py --efs 70 --nq 65536 --reps 3 2>&1 | grep -E "^ef="

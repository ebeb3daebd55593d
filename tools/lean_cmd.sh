# development aid: the lean kernel (keys-only result set) against the fast kernel across ef, single 10k launch and steady state
cd $GRAFT_REPO_ROOT
python tools/qbench.py --efs 70 > /dev/null 2>&1
echo "fast"; python tools/qbench.py --efs 32,70,96,128,192,256 --check 2>&1 | grep -E "^ef=|oracle"
echo "lean"; HS_LEAN_MIN_EF=1 python tools/qbench.py --efs 32,70,96,128,192,256 --check 2>&1 | grep -E "^ef=|oracle"
echo "fast steady"; python tools/qbench.py --efs 128,256 --nq 32768 --reps 3 2>&1 | grep "^ef="
echo "lean steady"; HS_LEAN_MIN_EF=1 python tools/qbench.py --efs 128,256 --nq 32768 --reps 3 2>&1 | grep "^ef="

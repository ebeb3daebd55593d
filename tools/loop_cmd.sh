# small launches: the visited set grows into the LDS nobody else needs
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 || exit 1
python tools/qbench.py --efs 70 > /dev/null 2>&1
for NQ in 256 1250 4096 10000; do echo "nq=$NQ"; python tools/qbench.py --efs 32,70,128,256 --nq $NQ --reps 5 --check 2>&1 | grep -E "^ef=|equal False"; done
EFS=256,384 python -u tools/other_configs.py gist 2>&1 | grep -E "^ef=|OPERATING"

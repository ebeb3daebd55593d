# development A/B: visited-set tier 1 in 16-bit slots (default) against the 32-bit form (HS_VIS16=0)
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 || exit 1
python tools/qbench.py --efs 70 > /dev/null 2>&1
for V in 0 1; do
  for NQ in 1250 10000; do echo "HS_VIS16=$V nq=$NQ"; HS_VIS16=$V python tools/qbench.py --efs 70,192,256 --nq $NQ --reps 4 --check 2>&1 | grep -E "^ef=|equal False"; done
done

# development A/B: flat start of the fast kernel (default) against the heap path from the first expansion (HS_FLAT=0)
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 || exit 1
python tools/qbench.py --efs 70 > /dev/null 2>&1
for NQ in 1250 10000 65536; do
  echo "nq=$NQ heap"; HS_FLAT=0 python tools/qbench.py --efs 32,70,128 --nq $NQ --reps 5 2>&1 | grep -E "^ef="
  echo "nq=$NQ flat"; HS_VERBOSE=1 python tools/qbench.py --efs 32,70,128 --nq $NQ --reps 5 --check 2>&1 | grep -E "^ef=|equal False"
done

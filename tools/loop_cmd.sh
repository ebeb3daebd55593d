# experiment: which property of the start order matters (HS_ORDER_KEY 0 far first, 1 near first, 2 by entry node id)
cd $GRAFT_REPO_ROOT
python tools/qbench.py --efs 70 > /dev/null 2>&1
for K in 0 1 2; do
  echo "HS_ORDER_KEY=$K"; HS_ORDER_KEY=$K python tools/qbench.py --efs 70 2>&1 | grep -E "^ef="
  HS_ORDER_KEY=$K python tools/qbench.py --efs 70 --nq 65536 --reps 3 2>&1 | grep -E "^ef="
done

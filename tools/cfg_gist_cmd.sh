cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for G in "4096,24,40,1.5" "2048,32,40,2.5"; do
  echo "=== GEN=$G"; GEN=$G EFS=64,128,256,384,512 python tools/other_configs.py gist 2>&1 | grep -v "amdgpu.ids"
done

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for G in "4096,16,40,1" "4096,24,40,1.5"; do
  echo "=== GEN=$G"; GEN=$G EFS=32,64,128,256 python tools/other_configs.py gist 2>&1 | grep -v "amdgpu.ids"
done

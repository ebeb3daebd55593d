set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/gather_calib.hip -o /tmp/gather_calib
for m in A B C; do /tmp/gather_calib $m 4096 256; done
/tmp/gather_calib A 8192 256; /tmp/gather_calib A 2048 512; /tmp/gather_calib A 16384 64; /tmp/gather_calib C 16384 64; /tmp/gather_calib A 1024 1024

# what a byte-moving kernel reaches on this box (copy, read, random row gathers with the search kernels' lane mappings)
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/hbm_ceiling.hip -o /tmp/hbm_ceiling || exit 1
timeout -k 10 300 /tmp/hbm_ceiling > gpurun_out/hbm_ceiling.log 2>&1
cat gpurun_out/hbm_ceiling.log

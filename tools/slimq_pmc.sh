# PMC instruction mix of the SlimQ kernels (run through gpurun): EFS=<ef> tools/slimq_pmc.sh
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/slimq_pmc
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/insts -- python tools/slimq_config.py sift > $O/run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/slimq_config.py sift > $O/run_kt.log 2>&1
find $O -name "*agent_info.csv" -delete
python - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/slimq_pmc/insts/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[k] += 1
    for k in agg:
        if "slimq" in k: print(k, n[k], {c: round(v / max(n[k], 1)) for c, v in agg[k].items()})
for f in glob.glob("gpurun_out/slimq_pmc/kt/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY

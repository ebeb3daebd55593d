# PMC passes over tools/qbench.py (development aid, run through gpurun): instruction mix and where the wave cycles go.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmc_group
rm -rf $O; mkdir -p $O
EF=${EF:-68}
python tools/qbench.py --efs $EF --reps 3 > $O/plain.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/insts -- python tools/qbench.py --efs $EF --reps 3 > $O/insts.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $O/wait -- python tools/qbench.py --efs $EF --reps 3 > $O/wait.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $O/lds -- python tools/qbench.py --efs $EF --reps 3 > $O/lds.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/qbench.py --efs $EF --reps 5 > $O/kt.log 2>&1
find $O -name "*agent_info.csv" -delete
python - <<'PY'
import csv, glob, collections
for d in ("insts", "wait", "lds"):
    for f in glob.glob(f"gpurun_out/pmc_group/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "hs::" not in k or "bf_" in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        for k in agg:
            print(d, k, {c: round(v / n[k][c]) for c, v in agg[k].items()}, "dispatches", max(n[k].values()))
for f in glob.glob("gpurun_out/pmc_group/kt/**/*kernel_stats.csv", recursive=True):
    for i, l in enumerate(open(f)):
        if i == 0 or "hs::" in l: print(l.strip()[:300])
PY
cat $O/plain.log | tail -2

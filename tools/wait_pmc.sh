# Where do the waves' cycles go? (run through gpurun)  WAIT_ANY = parked on s_waitcnt/barrier, WAIT_INST_ANY = issue stall,
# ACTIVE_INST_* = issuing; all in quad-cycles, disjoint, summing to WAVE_CYCLES (MI355X_MICROARCH.md SQ table).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/wait_pmc
mkdir -p $O
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"
python bench.py --index-dir /tmp/idx --ef 68 --streams 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc $C --output-format csv -d $O/fp32 -- python bench.py --index-dir /tmp/idx --ef 68 --streams 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/fp32.log 2>&1
EFS=256 rocprofv3 --pmc $C --output-format csv -d $O/slimq -- python tools/slimq_config.py sift > $O/slimq.log 2>&1
find $O -name "*agent_info.csv" -delete
python - <<'PY'
import csv, glob, collections
for d in ("fp32", "slimq"):
    for f in glob.glob(f"gpurun_out/wait_pmc/{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            if "hs::" not in k: continue
            if int(r["Grid_Size"]) < 64 * 5000: continue      # the main launches only
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
        for k in agg:
            w = agg[k]["SQ_WAVE_CYCLES"]
            print(d, k, n[k], {c: f"{v / w * 100:.1f}%" for c, v in agg[k].items() if c != "SQ_WAVE_CYCLES"}, "wave quad-cycles per launch", round(w / max(n[k], 1)))
PY

#!/bin/bash
# round 3: GIST-1M-like (d = 960, 3.8 GB of rows) with 10k-query launches instead of configs[2]'s 1k: what the long-row kernel
# reaches when the launch fills the chip; FETCH_SIZE in its own pass
export TMPDIR=/tmp IDX_DIR=/tmp/gist_big1m NQ=10000
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_gist1m_bigbatch
rm -rf $O; mkdir -p $O
EFS=64,256,384,512 python -u tools/other_configs.py gist 1000000 > $O/sweep.log 2>&1 || { tail -5 $O/sweep.log; kill $HB; exit 1; }
export PROFILE_EF=384
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/other_configs.py gist 1000000 > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python tools/other_configs.py gist 1000000 > /dev/null 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
kill $HB
grep "^ef=\|^build\|OPERATING" $O/sweep.log | cut -c1-220
python - <<PY
import csv, glob
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for l in open(f):
        if "hs::" in l: print(l.strip()[:200])
for f in glob.glob("$O/pmc_FETCH_SIZE/**/*counter_collection.csv", recursive=True):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if "flat_kernel" in r["Kernel_Name"]: tot += float(r["Counter_Value"]); n += 1
    print("FETCH_SIZE flat_kernel KiB per dispatch", round(tot / max(n, 1)), "over", n, "dispatches")
PY

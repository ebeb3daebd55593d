#!/bin/bash
# round 3: one of the other BASELINE.json shapes (gist | deep): sweep against the oracle, then the operating point under rocprofv3
# (kernel trace, and FETCH_SIZE / WRITE_SIZE in their own --pmc passes)   usage: r03_cfg_cmd.sh gist|deep [n]
W=$1; N=$2
export TMPDIR=/tmp
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_cfg_$W
rm -rf $O; mkdir -p $O
export IDX_DIR=/tmp/cfg_$W
python -u tools/other_configs.py $W $N > $O/sweep.log 2>&1 || { tail -5 $O/sweep.log; kill $HB; exit 1; }
cat $O/sweep.log | grep -v "^ef=.*oracle_match_first200=True" | tail -5; grep -c "oracle_match_first200=True" $O/sweep.log
EF=$(cat $IDX_DIR/operating_ef 2>/dev/null || echo 256)
export PROFILE_EF=$EF
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/other_configs.py $W $N > $O/kt.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d $O/pmc_$C -- python tools/other_configs.py $W $N > /dev/null 2>&1
done
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
kill $HB
python - <<PY
import csv, glob
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for l in open(f):
        if "hs::" in l or l.startswith('"Name"'): print(l.strip()[:200])
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        tot, n = {}, {}
        for r in csv.DictReader(open(f)):
            if "hs::" in r["Kernel_Name"] and r["Counter_Name"] == c:
                k = r["Kernel_Name"].split("(")[0][:60]
                tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
        for k in tot: print(c, k, "KiB per dispatch %.0f over %d dispatches" % (tot[k] / n[k], n[k]))
PY

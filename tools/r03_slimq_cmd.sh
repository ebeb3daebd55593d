#!/bin/bash
# round 3: HNSW-SlimQ at 1M (sift: d=128 L2 on the bench's data; cohere: d=768 IP, configs[4]): sweep against the oracle, then the
# ef=256 point under rocprofv3 (kernel trace; FETCH_SIZE and the instruction mix in their own --pmc passes)   usage: r03_slimq_cmd.sh sift|cohere [rq]
W=$1
export GRAPH=${2:-hnswlib}
export TMPDIR=/tmp
( while true; do sleep 60; date >> gpurun_out/heartbeat.log; done ) &
HB=$!
O=gpurun_out/r03_slimq_${W}_$GRAPH
rm -rf $O; mkdir -p $O
export IDX_DIR=/tmp/slimq_${W}_$GRAPH
if [ "$W" = "sift" ]; then export EFS=64,128,256,384; else export EFS=64,256,1024; fi
python -u tools/slimq_config.py $W > $O/sweep.log 2>&1 || { tail -5 $O/sweep.log; kill $HB; exit 1; }
grep "^ef=\|^build\|^graph\|prep" $O/sweep.log | cut -c1-260
export PROFILE_EF=256
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python tools/slimq_config.py $W > $O/kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_FETCH_SIZE -- python tools/slimq_config.py $W > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_insts -- python tools/slimq_config.py $W > /dev/null 2>&1
find $O -name "*agent_info.csv" -delete; find $O -name "*kernel_trace.csv" -delete
kill $HB
python - <<PY
import csv, glob
for f in glob.glob("$O/kt/**/*kernel_stats.csv", recursive=True):
    for l in open(f):
        if "hs::" in l or l.startswith('"Name"'): print(l.strip()[:220])
for d in ("pmc_FETCH_SIZE", "pmc_insts"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % d, recursive=True):
        tot, n = {}, {}
        for r in csv.DictReader(open(f)):
            if "slimq_kernel" in r["Kernel_Name"]:
                k = r["Counter_Name"]
                tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"]); n[k] = n.get(k, 0) + 1
        print(d, "slimq_kernel per dispatch:", {k: round(tot[k] / n[k]) for k in tot}, "dispatches", max(n.values()) if n else 0)
PY

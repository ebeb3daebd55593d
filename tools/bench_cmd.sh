# the bench line as the driver runs it + the two-rank rehearsals (both ranks on cuda:0, gloo) of the weak and strong modes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2_bench
mkdir -p $O
python bench.py --index-dir /tmp/idx > $O/bench_1gpu.json 2> $O/bench_1gpu.log || { tail -20 $O/bench_1gpu.log; exit 1; }
for M in weak strong; do
  HS_BENCH_ONE_DEVICE=1 HS_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 5 --warmup 1 --ef 68 --index-dir /tmp/idx --scaling $M --no-cpu-baseline > $O/bench_2rank_$M.json 2> $O/bench_2rank_$M.log || { tail -20 $O/bench_2rank_$M.log; exit 1; }
done
python - <<'PY'
import json
for f in ("bench_1gpu", "bench_2rank_weak", "bench_2rank_strong"):
    j = json.loads(open(f"gpurun_out/r2_bench/{f}.json").read().strip().split("\n")[-1])   # (gloo prints connection chatter on stdout)
    c = j["config"]
    print(f, "value", j["value"], "scaling", j["scaling"], "ms/step", j["ms_per_step"], "ef", c["ef_search"], "recall", c["recall_at_10"], "frac", j["roofline"]["frac"],
          "launch_ms", j["roofline"]["launch_ms"], "dev-resident", c["device_resident_pipelined_qps"], "sync-host", c["sync_host_pointer_api_qps_pageable"], "cpu", (j["cpu_baseline"] or {}).get("value"))
PY

#!/bin/bash
# round 3: whole GPU suite, the bench line as the driver runs it, and the two-rank rehearsal (both ranks on cuda:0, gloo)
mkdir -p gpurun_out
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/r03_t3.log 2>&1; rc=$?; tail -3 $O/r03_t3.log
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 900 python bench.py --index-dir /tmp/hsidx --steps 20 --warmup 5 > $O/r03_bench_1gpu.json 2> $O/r03_bench_1gpu.log || { tail -20 $O/r03_bench_1gpu.log; exit 1; }
HS_BENCH_ONE_DEVICE=1 HS_BENCH_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 2 --warmup 1 --ef 68 --index-dir /tmp/hsidx --no-cpu-baseline > $O/r03_bench_2rank.json 2> $O/r03_bench_2rank.log || { tail -20 $O/r03_bench_2rank.log; exit 1; }
python - <<'PY'
import json
for f in ("r03_bench_1gpu", "r03_bench_2rank"):
    j = json.loads(open(f"gpurun_out/{f}.json").read().strip().split("\n")[-1])
    c = j["config"]
    print(f, "value", j["value"], "scaling", j["scaling"], "ms/step", j["ms_per_step"], "timed_s", c["timed_seconds"], "ef", c["ef_search"], "recall", c["recall_at_10"], "frac", j["roofline"]["frac"],
          "launch_ms", j["roofline"]["launch_ms"], "kernel", j["roofline"]["kernel"], "dev-resident", c["device_resident_pipelined_qps"], "weak side", c["weak_scaling_qps_side_figure"], "cpu", j["cpu_baseline"])
PY

# same-box A/B of the judged bench: lean kernel from ef=64 (default) against the fast kernel (HS_LEAN_MIN_EF=100000)
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 || exit 1
run() {
  python bench.py --index-dir /tmp/idx --ef 70 --no-cpu-baseline > gpurun_out/ab_$1.json 2> gpurun_out/ab_$1.log
  python - "$1" <<PY
import json, sys
j=json.loads(open(f"gpurun_out/ab_{sys.argv[1]}.json").read().strip().split("\n")[-1]); c=j["config"]; r=j["roofline"]
print(sys.argv[1], "value",j["value"],"frac",r["frac"],"launch_ms",r["launch_ms"],"timed",r["timed_region_frac"],"devres",c["device_resident_pipelined_qps"])
PY
}
for rep in 1 2; do HS_LEAN_MIN_EF=100000 run fast; run lean; done

#!/usr/bin/env python3
"""Turn the raw output of tools/prof_cmd.sh (gpurun_out/r01) into the committed summaries under profiles/:
bench JSON lines, rocprofv3 kernel stats, and the HBM-traffic figure bench.py reports as roofline.traffic."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r01")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def newest(pattern):
    f = sorted(glob.glob(os.path.join(SRC, pattern), recursive=True), key=os.path.getmtime)
    return f[-1] if f else None


def per_launch(path, kernel_sub, counters):
    tot = {c: 0.0 for c in counters}
    n = {c: 0 for c in counters}
    for r in csv.DictReader(open(path)):
        if kernel_sub in r["Kernel_Name"] and r["Counter_Name"] in tot:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
    return {c: tot[c] / max(n[c], 1) * DISPATCHES_PER_PASS.get(kernel_sub, 1) for c in counters}, n


# A 10k-query pass of the fast kernel is two dispatches since round 2 (descent, then level-0 search in entry-distance order,
# csrc/capi.cpp search_dev_group): "per launch" figures are per PASS, i.e. the per-dispatch average times two.
DISPATCHES_PER_PASS = {"fast_kernel": 2, "lean_kernel": 2, "flat_kernel": 2} if (len(sys.argv) > 1 and sys.argv[1] != "r01") else {}


for s in ("1stream", "pipelined"):
    shutil.copy(os.path.join(SRC, f"bench_{s}.json"), os.path.join(DST, f"{tag}_bench_1gpu_{s}.json"))
    os.makedirs(DST, exist_ok=True)
    ks = newest(f"kt_{s}/**/*kernel_stats.csv")
    rows = [l for i, l in enumerate(open(ks)) if i == 0 or "hs::" in l]
    open(os.path.join(DST, f"{tag}_kernel_stats_{s}.csv"), "w").writelines(rows)
b = json.load(open(os.path.join(SRC, "bench_1stream.json")))
ef = b["config"]["ef_search"]
KERNEL = b["roofline"].get("kernel", "hs::fast_kernel").split("::")[-1]   # the search kernel that served the bench shape
fetch, _ = per_launch(newest("pmc_FETCH_SIZE/**/*counter_collection.csv"), KERNEL, ["FETCH_SIZE"])
write, _ = per_launch(newest("pmc_WRITE_SIZE/**/*counter_collection.csv"), KERNEL, ["WRITE_SIZE"])
tcc, _ = per_launch(newest("pmc_TCC_HIT_sum/**/*counter_collection.csv"), KERNEL, ["TCC_HIT_sum", "TCC_MISS_sum"])
cal, ncal = per_launch(newest("pmc_calib/**/*counter_collection.csv"), "gather", ["FETCH_SIZE"])
known = 1 << 30
corr = known / (cal["FETCH_SIZE"] * 1024)
hbm = fetch["FETCH_SIZE"] * 1024 * corr + write["WRITE_SIZE"] * 1024
out = {
    "workload": f"SIFT-1M-like d=128 N=1000000 nq=10000 ef={ef} k=10, 1 stream",
    "ef": ef,
    "kernel": "hs::" + KERNEL,
    "kernel_note": "both dispatches of a pass: descent + level-0 search" if DISPATCHES_PER_PASS else "",
    "FETCH_SIZE_KiB_per_launch": round(fetch["FETCH_SIZE"], 1),
    "WRITE_SIZE_KiB_per_launch": round(write["WRITE_SIZE"], 1),
    "calibration": {
        "kernel": "tools/gather_calib.hip mode A (same lane mapping as wave_dists), 1 GiB of distinct random 512-B rows, average of the launches",
        "known_bytes": known, "FETCH_SIZE_KiB": round(cal["FETCH_SIZE"], 1), "fetch_correction": round(corr, 4)},
    "hbm_bytes_per_launch": int(hbm),
    "TCC_HIT_sum": int(tcc["TCC_HIT_sum"]), "TCC_MISS_sum": int(tcc["TCC_MISS_sum"]),
    "algorithmic_bytes_per_launch": b["roofline"].get("algorithmic_bytes_per_launch", b["roofline"].get("algorithmic_bytes_per_step")),
    "note": "FETCH_SIZE on gfx950 counts 64 B per 128-B request (MI355X_MICROARCH.md HBM); corrected by the factor measured on "
            "this access pattern; separate --pmc passes for FETCH_SIZE, WRITE_SIZE and TCC_HIT/MISS (tools/prof_cmd.sh)"}
json.dump(out, open(os.path.join(DST, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))


def totals(path, kernel_sub):
    tot, disp = {}, set()
    for r in csv.DictReader(open(path)):
        if kernel_sub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    return tot, len(disp)


# instruction mix and wait split of the search kernel (separate --pmc passes), per PASS of 10k queries
lines = []
for d, title in (("pmc_insts", "instruction mix"), ("pmc_wait", "where a wavefront's cycles go")):
    f = newest(f"{d}/**/*counter_collection.csv")
    if not f:
        continue
    tot, nd = totals(f, KERNEL)
    passes = max(nd // DISPATCHES_PER_PASS.get(KERNEL, 1), 1)
    lines.append(f"{title}: rocprofv3 --pmc {' '.join(sorted(tot))} -- python bench.py --ef {ef} --streams 1 --steps 2 --warmup 1 "
                 f"(hs::{KERNEL}, {nd} dispatches = {passes} passes of 10000 queries; per-pass averages; SQ_WAVE_CYCLES and the wait/active counters count quad-cycles)")
    for k in sorted(tot):
        lines.append(f"    {k:22s} {tot[k] / passes:16.0f}")
    if "SQ_INSTS_VALU" in tot:
        nd_q = b["config"]["sweep"][str(ef)]["n_dist"]
        lines.append(f"    -> per query: {tot['SQ_INSTS_VALU'] / passes / 1e4:.0f} VALU + {tot['SQ_INSTS_SALU'] / passes / 1e4:.0f} SALU + {tot['SQ_INSTS_LDS'] / passes / 1e4:.0f} LDS; "
                     f"per distance evaluation ({nd_q:.0f} per query): {tot['SQ_INSTS_VALU'] / passes / 1e4 / nd_q:.0f} VALU + {tot['SQ_INSTS_SALU'] / passes / 1e4 / nd_q:.0f} SALU")
    if "SQ_WAIT_ANY" in tot:
        wc = tot["SQ_WAVE_CYCLES"]
        lines.append("    -> share of wave cycles: " + ", ".join(f"{k} {100 * tot[k] / wc:.1f}%" for k in sorted(tot) if k != "SQ_WAVE_CYCLES"))
if lines:
    open(os.path.join(DST, f"{tag}_pmc_insts.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
cv = os.path.join(SRC, "convert_1m.log")
if os.path.exists(cv):
    shutil.copy(cv, os.path.join(DST, f"{tag}_convert_1m.log"))

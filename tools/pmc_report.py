"""Fold the rocprofv3 passes of tools/profile_round.sh into profiles/<round>/:
   kernel_stats.csv   -- per-kernel calls / total / average duration (the --stats pass, HIP-graph replay)
   pmc_traffic.json   -- HBM bytes per launch per kernel (2 x FETCH_SIZE KB gfx950 correction + WRITE_SIZE KB)
   pmc_counters.json  -- matrix-core / occupancy / LDS / L2 counters per launch for the top kernels + derived figures
python tools/pmc_report.py gpurun_out/<dir> profiles/r02"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", n).split("(")[0]


def counters(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return acc


stats = glob.glob(src + "/stats/**/*kernel_stats.csv", recursive=True)
rows = []
if stats:
    shutil.copy(stats[0], os.path.join(dst, "bench_n1_graph_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
dur = {short(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6) for r in rows}

allc = defaultdict(dict)
for d in sorted(glob.glob(src + "/pmc*/")):
    for k, cs in counters(d).items():
        for c, vals in cs.items():
            allc[k][c] = (sum(vals) / len(vals), len(vals))

traffic = {"method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, python3 bench.py --steps 2 "
                     "--warmup 1 --no-cpu-baseline --no-ade --no-graph on one MI355X; per-dispatch values averaged per kernel "
                     "symbol; FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md", "kernels": {}}
for k, cs in allc.items():
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        f, w = cs["FETCH_SIZE"][0], cs["WRITE_SIZE"][0]
        traffic["kernels"][k] = {"launches_sampled": cs["FETCH_SIZE"][1], "fetch_kb_raw_avg": f, "write_kb_avg": w,
                                 "hbm_bytes_per_launch": (2 * f + w) * 1024}
json.dump(traffic, open(os.path.join(dst, "pmc_traffic.json"), "w"), indent=1)

top = sorted(dur, key=lambda k: -dur[k][2])[:14]
rep = {"method": "tools/profile_round.sh: one rocprofv3 --pmc pass per counter group (kernel-trace only, eager launches); values "
                 "are per-dispatch averages per kernel symbol.  avg_us / calls / total_ms from the --stats pass of the "
                 "graph-replayed bench.  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x SQ_BUSY_CU_CYCLES); "
                 "occupancy_waves_per_cu = SQ_WAVE_CYCLES / SQ_BUSY_CU_CYCLES (quad-cycle units cancel); lds_conflict_frac = "
                 "SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS).", "kernels": {}}
for k in top:
    cs = {c: v[0] for c, v in allc.get(k, {}).items()}
    e = {"calls": dur[k][0], "avg_us": round(dur[k][1], 2), "total_ms": round(dur[k][2], 3), "counters_per_launch": cs}
    if cs.get("SQ_BUSY_CU_CYCLES"):
        if "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
            e["mfma_busy_frac"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * cs["SQ_BUSY_CU_CYCLES"])
    if cs.get("SQ_BUSY_CYCLES") and "SQ_WAVE_CYCLES" in cs:
        e["occupancy_waves_per_se_cycle"] = cs["SQ_WAVE_CYCLES"] / cs["SQ_BUSY_CYCLES"]
    if cs.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_conflict_frac"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
    if cs.get("TCC_HIT_sum") is not None and cs.get("TCC_MISS_sum") is not None and cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"] > 0:
        e["l2_hit"] = cs["TCC_HIT_sum"] / (cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"])
    if k in traffic["kernels"]:
        e["hbm_bytes_per_launch"] = traffic["kernels"][k]["hbm_bytes_per_launch"]
    rep["kernels"][k] = e
json.dump(rep, open(os.path.join(dst, "pmc_counters.json"), "w"), indent=1)
for k in top:
    e = rep["kernels"][k]
    print(f"{k[:52]:52s} {e['avg_us']:8.1f} us x{e['calls']:5d}  mfma {e.get('mfma_busy_frac', float('nan')):.3f}  "
          f"lds-conf {e.get('lds_conflict_frac', float('nan')):.3f}  l2 {e.get('l2_hit', float('nan')):.2f}  "
          f"hbm {e.get('hbm_bytes_per_launch', 0) / 1e6:8.2f} MB")

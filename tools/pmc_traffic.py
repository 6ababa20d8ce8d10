"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) into profiles/r01/pmc_traffic.json:
per kernel symbol, average HBM bytes per launch = 2 * FETCH_SIZE KB (gfx950 correction, MI355X_MICROARCH.md)
+ WRITE_SIZE KB.   python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import csv, glob, json, re, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
                n = re.sub(r"^void ", "", n).split("(")[0]
                acc[n].append(float(r["Counter_Value"]))
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (kernel-trace only), python3 bench.py "
                 "--steps 2 --warmup 1 --no-cpu-baseline --no-graph on one MI355X; per-dispatch values averaged per kernel "
                 "symbol; FETCH_SIZE x2 per the gfx950 correction in MI355X_MICROARCH.md (sanity: adamw_clip_kernel reads "
                 "4 and writes 3 fp32 arrays of the 75.3 M parameters)",
       "kernels": {}}
for k in sorted(set(fetch) & set(write), key=lambda k: -sum(fetch[k])):
    f, w = sum(fetch[k]) / len(fetch[k]), sum(write[k]) / len(write[k])
    out["kernels"][k] = {"launches_sampled": len(fetch[k]), "fetch_kb_raw_avg": f, "write_kb_avg": w,
                         "hbm_bytes_per_launch": (2 * f + w) * 1024}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernels:", len(out["kernels"]))
for k in list(out["kernels"])[:12]:
    print(f"{k:45s} {out['kernels'][k]['hbm_bytes_per_launch'] / 1e6:10.2f} MB/launch  ({out['kernels'][k]['launches_sampled']} launches)")

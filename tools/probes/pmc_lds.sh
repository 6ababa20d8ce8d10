# LDS bank-conflict counters of the default bench (eager launches):  pmc_lds.sh OUTDIR [PATTERN]
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; pat=${2:-attn}
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc -o pmc -- python3 $ROOT/bench.py --no-cpu-baseline --no-ade --steps 2 --warmup 1 --no-graph > $out/pmc.log 2>&1 || { echo failed; tail -3 $out/pmc.log; exit 1; }
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if not any(p in n for p in "$pat".split("|")): continue
    import re
    k = re.sub(r"\(anonymous namespace\)::|^void ", "", n).split("(")[0][:50]
    if r["Counter_Name"] == "SQ_LDS_BANK_CONFLICT": acc[k][0] += float(r["Counter_Value"]); acc[k][2] += 1
    elif r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE": acc[k][1] += float(r["Counter_Value"])
for k, (c, a, n) in sorted(acc.items()):
    print(f"{k:52s} launches {n:5d}  bank_conflict / idx_active = {c / max(a, 1):.3f}")
PY
find $out/pmc -name "*.csv" -size +5M -delete

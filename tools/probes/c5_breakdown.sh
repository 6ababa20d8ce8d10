# C5: trunk-cached vs look-ahead-trunk step, and the per-kernel statistics of the C5 step
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$1; mkdir -p $out
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
line() { name=$1; shift; timeout -k 10 300 python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10 "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; exit 1; }; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
line c5 --case C5 && line c5_cached --case C5 --trunk-cache && line c5_b --case C5 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-ade --case C5 --steps 10 --warmup 3 > "$out/stats.log" 2>&1 || { echo "stats failed"; exit 1; }
find "$out" -name "*agent_info.csv" -delete
python3 - <<PY
import csv, glob, collections
f = glob.glob("$out/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:28]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"][:9].rjust(10), r["Percentage"][:5])
PY

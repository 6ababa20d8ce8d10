# rocprofv3 kernel statistics of the default bench under two settings (quoted "VAR=VALUE ..." strings); prints the kernels matching PATTERN
#   stats_ab.sh OUTDIR "SETTING_A" "SETTING_B" PATTERN
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; pat=$4
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
i=0
for setting in "$2" "$3"; do
  i=$((i + 1))
  for kv in $setting; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/s$i -o k -- python3 $ROOT/bench.py --no-cpu-baseline --no-ade --steps 10 --warmup 3 > $out/s$i.log 2>&1 || { echo "run $i failed"; tail -3 $out/s$i.log; exit 1; }
  for kv in $setting; do unset "${kv%%=*}"; done
  echo "== $setting"
  python3 - <<PY
import csv, glob
f = glob.glob("$out/s$i/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(p in r["Name"] for p in "$pat".split("|")):
        print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), r["AverageNs"][:8].rjust(9), r["TotalDurationNs"].rjust(11))
PY
  find $out/s$i -name "*trace.csv" -delete
done

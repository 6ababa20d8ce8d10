# kernel trace of one bench case -> step_trace:  trace_case.sh OUTDIR [bench args...]
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; shift
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $out/t -o k -- python3 $ROOT/bench.py --no-cpu-baseline --no-ade --steps 6 --warmup 3 "$@" > $out/t.log 2>&1 || { echo "trace failed"; tail -3 $out/t.log; exit 1; }
python3 $ROOT/tools/step_trace.py $(find $out/t -name "*kernel_trace.csv") > $out/step.txt || exit 1
find $out/t -name "*.csv" -delete
head -1 $out/step.txt

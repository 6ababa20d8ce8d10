# kernel traces of the default bench under two settings of one environment variable -> step_trace of each
#   trace_ab.sh OUTDIR VAR A B
out=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/$1; ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; var=$2
mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for v in $3 $4; do
  export $var=$v
  rocprofv3 --kernel-trace --output-format csv -d $out/t$v -o k -- python3 $ROOT/bench.py --no-cpu-baseline --no-ade --steps 6 --warmup 3 > $out/t$v.log 2>&1 || { echo "trace $v failed"; tail -3 $out/t$v.log; exit 1; }
  python3 $ROOT/tools/step_trace.py $(find $out/t$v -name "*kernel_trace.csv") > $out/step_$v.txt || exit 1
  head -1 $out/step_$v.txt
done

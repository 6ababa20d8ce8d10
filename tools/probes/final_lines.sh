# round-end record, part 1: rocprofv3 passes of the default bench + the default bench lines kept under profiles/
#   final_lines.sh OUTDIR [part]      part 1 (default): profile + default lines; part 2: the other configurations
out=gpurun_out/$1; part=${2:-1}
mkdir -p $out
line() { name=$1; shift; timeout -k 10 500 python bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; exit 1; }; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel'][:40], round(d['roofline']['frac'],3))"; }
if [ "$part" = "1" ]; then
  bash tools/profile_round.sh $1/prof 2>&1 | tail -8 || exit 1
  cd ${GRAFT_REPO_ROOT:-$(pwd)}
  line bench_n1_default && line bench_n1_default_20 --steps 20 --warmup 5
else
  line bench_n1_c4 --case C4 --no-ade --no-cpu-baseline && line bench_n1_c5 --case C5 --no-ade --no-cpu-baseline \
    && line bench_n1_paper_dropout --dropout paper --no-ade --no-cpu-baseline && line bench_n1_trunk_cached --trunk-cache --no-ade --no-cpu-baseline \
    && line bench_n1_c5_trunk_cached --case C5 --trunk-cache --no-ade --no-cpu-baseline && line bench_n1_default_20_noade --steps 20 --warmup 5 --no-ade --no-cpu-baseline
fi

# round-end record: rocprofv3 passes of the default bench + the bench lines kept under profiles/
out=gpurun_out/$1
mkdir -p $out
bash tools/profile_round.sh $1/prof 2>&1 | tail -8 || exit 1
cd ${GRAFT_REPO_ROOT:-$(pwd)}
line() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; exit 1; }; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3), d['roofline']['kernel'][:40], round(d['roofline']['frac'],3))"; }
line bench_n1_default && line bench_n1_default_20 --steps 20 --warmup 5 --no-ade --no-cpu-baseline && line bench_n1_c4 --case C4 && line bench_n1_c5 --case C5 \
  && line bench_n1_paper_dropout --dropout paper --no-ade --no-cpu-baseline && line bench_n1_trunk_cached --trunk-cache --no-ade --no-cpu-baseline

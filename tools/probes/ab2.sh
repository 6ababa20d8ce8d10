# alternating 60-step bench runs of two environment settings (each a quoted "VAR=VALUE VAR=VALUE" string), N pairs
#   [BENCH_ARGS="--case C5"] ab2.sh OUTDIR "SETTING_A" "SETTING_B" [PAIRS]
out=gpurun_out/$1; a=$2; b=$3; n=${4:-3}
mkdir -p $out
B="python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10 $BENCH_ARGS"
run() { name=$1; shift; env $@ $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in $(seq 1 $n); do run a$i $a && run b$i $b || exit 1; done

# alternating 60-step bench runs of two settings of one environment variable:  ab_env.sh OUTDIR VAR VALUE_A VALUE_B
out=gpurun_out/$1; var=$2; a=$3; b=$4
mkdir -p $out
B="python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10"
run() { name=$1; shift; env "$@" $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
run a1 $var=$a && run b1 $var=$b && run a2 $var=$a && run b2 $var=$b

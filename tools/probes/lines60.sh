# 60-step lines of the bench configurations (no CPU legs):  lines60.sh OUTDIR
out=gpurun_out/$1; mkdir -p $out
line() { name=$1; shift; timeout -k 10 300 python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10 "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; exit 1; }; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
line c2 && line c4 --case C4 && line c5 --case C5 && line paper --dropout paper && line cached --trunk-cache && line c2_again

# A/B of one environment switch on the box: a focused test subset first, then alternating 60-step bench runs.
#   usage: tools/probes/ab.sh OUTDIR SWITCH "PYTEST -k EXPRESSION"
out=gpurun_out/$1; sw=$2; sel=$3
mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_widen.py -m gpu -x -q -k "$sel" > $out/t.log 2>&1
echo "pytest rc=$? segv=$(grep -c Segmentation $out/t.log)"; tail -1 $out/t.log | cut -c1-100
grep -E "^(FAILED|ERROR)|Error|assert " $out/t.log | head -20
B="python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10"
run() { name=$1; shift; env "$@" $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
run off1 $sw=0 && run on1 $sw=1 && run off2 $sw=0 && run on2 $sw=1

# focused tests + copy-site report + two bench runs (no A/B):  tools/probes/run1.sh OUTDIR "PYTEST -k EXPRESSION"
out=gpurun_out/$1; sel=$2
mkdir -p $out
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_widen.py -m gpu -x -q -k "$sel" > $out/t.log 2>&1
echo "pytest rc=$? segv=$(grep -c Segmentation $out/t.log)"; tail -1 $out/t.log | cut -c1-100
grep -E "^(FAILED|ERROR)|Error" $out/t.log | head -20
timeout -k 10 200 python tools/copy_sites.py > $out/copy_sites.txt 2> $out/copy_sites.err; head -40 $out/copy_sites.txt
B="python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10"
run() { name=$1; shift; env "$@" $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
run a1 A=1 && run a2 A=1

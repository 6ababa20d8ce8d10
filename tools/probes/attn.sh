# attention phase probes (GPS encoder / fusion encoder / GPS decoder shapes), attention tests, two bench runs
out=gpurun_out/$1
mkdir -p $out
for shp in "8 8 40 104 1" "8 8 160 16 1" "8 8 70 104 2"; do
  timeout -k 10 150 python tools/attn_phase_probe.py $shp 2>&1 | grep -v -i "warning\|amdgpu.ids" > "$out/probe_$(echo $shp | tr ' ' '_').txt" || exit 1
done
cat $out/probe_*.txt
timeout -k 10 500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "$2" > $out/t.log 2>&1
echo "pytest rc=$? segv=$(grep -c Segmentation $out/t.log)"; tail -1 $out/t.log | cut -c1-100
grep -E "^(FAILED|ERROR)|Error" $out/t.log | head -20
B="python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10"
run() { name=$1; shift; env "$@" $B > $out/$name.json 2> $out/$name.err; python -c "import json; d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
run a1 A=1 && run a2 A=1

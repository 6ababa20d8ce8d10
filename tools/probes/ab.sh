mkdir -p gpurun_out/r4d
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -x -q -k "not full_size" > gpurun_out/r4d/t.log 2>&1; tail -3 gpurun_out/r4d/t.log
B="python bench.py --no-ade --no-cpu-baseline"
run() { name=$1; shift; env "$@" $B > gpurun_out/r4d/$name.json 2> gpurun_out/r4d/$name.err; python -c "import json; d=json.loads(open('gpurun_out/r4d/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
run base A=1 && run wside RF_WGRAD_SIDE=1 && run base2 A=1 && run wside2 RF_WGRAD_SIDE=1

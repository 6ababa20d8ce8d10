mkdir -p gpurun_out/r5s
timeout -k 10 900 python -m pytest tests/test_gpu_model.py tests/test_widen.py tests/test_gpu_kernels.py -m gpu -x -q -k "informer or train_step or widen or transformer or blocks or gemm or layernorm or layer_norm or dropout" > gpurun_out/r5s/t.log 2>&1; echo "segv=$(grep -c Segmentation gpurun_out/r5s/t.log)"; tail -1 gpurun_out/r5s/t.log | cut -c1-80
B="python bench.py --no-ade --no-cpu-baseline --steps 60 --warmup 10"
run() { name=$1; shift; env "$@" $B > gpurun_out/r5s/$name.json 2> gpurun_out/r5s/$name.err; python -c "import json; d=json.loads(open('gpurun_out/r5s/$name.json').read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],3))"; }
run off1 A=1 && run on1 RF_WGRAD_SIDE=1 && run off2 A=1 && run on2 RF_WGRAD_SIDE=1

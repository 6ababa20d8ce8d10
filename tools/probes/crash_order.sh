# The one pytest order that showed round 3's host segmentation fault inside hipGraphLaunch (full-size engine tests, then the
# dropout-variant engine test), once, in its own process, with a native backtrace handler (tools/probes/segv_bt.c).
#   crash_order.sh OUTDIR ["ENV=... ENV=..."]
out=gpurun_out/$1; mkdir -p $out
env RF_SEGV_BT=1 RF_SEGV_FILE=$out/segv.txt $2 timeout -k 10 600 python -m pytest tests/test_gpu_model.py \
  -k "full_size_train_step or graphed_engine_with_dropouts" -x -q -p no:cacheprovider > $out/order.log 2>&1
rc=$?
echo "rc $rc: $(grep -E 'passed|failed' $out/order.log)" | tee -a $out/summary.txt
exit $rc

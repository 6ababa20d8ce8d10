# The round-3 host segmentation fault inside hipGraphLaunch: the one pytest order that showed it, once per
# setting, each in its own process with a native backtrace handler.   crash_bisect.sh OUTDIR "ENV..." "ENV..." ...
# A host crash (rc 139 / 134) is the expected outcome of some settings and does not stop the chain; a time-out (124 / 137) does.
out=gpurun_out/$1; shift
mkdir -p $out
SEL='full_size_train_step or graphed_engine_with_dropouts'
i=0
for setting in "$@"; do
  i=$((i+1))
  echo "== run $i: $setting" | tee -a $out/summary.txt
  env RF_SEGV_BT=1 RF_ENGINE_DEBUG=1 $setting timeout -k 10 420 python -m pytest tests/test_gpu_model.py -k "${PYSEL:-$SEL}" -x -q -p no:cacheprovider > $out/run$i.log 2>&1
  rc=$?
  echo "   rc $rc: $(grep -E 'passed|failed|segv_bt' $out/run$i.log | tr '\n' ' ')" | tee -a $out/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping" | tee -a $out/summary.txt; exit 1; fi
done
exit 0

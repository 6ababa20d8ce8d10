# capture_probe under the system HIP runtime (ROCm 7.2) and under the one PyTorch ships (7.0.2, the one the engine runs on)
out=gpurun_out/$1; mkdir -p $out
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
for rt in ${RTS:-system torch}; do
  for mode in ${MODES:-0 1 2 3 4 5 7}; do
    if [ $rt = torch ]; then pre="env LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH"; else pre=""; fi
    $pre timeout -k 5 120 tools/probes/capture_probe ${ROUNDS:-300} $mode > $out/${rt}_m$mode.log 2>&1
    rc=$?
    echo "$rt runtime, NO_DESTROY=${NO_DESTROY:-}, mode $mode: rc $rc: $(tail -1 $out/${rt}_m$mode.log)" | tee -a $out/summary.txt
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping" | tee -a $out/summary.txt; exit 1; fi
  done
done
exit 0

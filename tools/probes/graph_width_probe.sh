# graph_width_probe over widths / destroy patterns / launch streams, under the HIP runtime PyTorch ships and the system one
out=gpurun_out/$1; mkdir -p $out
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
for rt in ${RTS:-torch system}; do
  if [ $rt = torch ]; then pre="env LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH"; else pre=""; fi
  for destroy in 0 1; do for created in 0 1; do for w in 1 2 3 4 5 6 7 8; do
    res=$($pre timeout -k 5 60 tools/probes/graph_width_probe $w ${ROUNDS:-200} $destroy $created 2>&1 | tail -1)
    rc=$?
    echo "$rt destroy=$destroy created=$created width=$w: $res" | tee -a $out/summary.txt
    case "$res" in *"Terminated"*|"") echo "no result: stopping"; exit 1;; esac
  done; done; done
done
exit 0

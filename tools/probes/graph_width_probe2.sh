# second sweep (torch's runtime only): which launch crashes, and does re-filling the destroyed exec's stream slots help?
out=gpurun_out/$1; mkdir -p $out
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
pre="env LD_PRELOAD=$TL/libamdhip64.so LD_LIBRARY_PATH=$TL:$LD_LIBRARY_PATH"
for cfg in "0 0" "0 1" "0 2" "-1 0" "1 0" "2 0" "4 0" "8 0"; do for w in 2 3 5 7; do
  res=$($pre timeout -k 5 90 tools/probes/graph_width_probe $w ${ROUNDS:-400} 1 0 $cfg 2>&1 | tail -1)
  echo "pad/which=$cfg width=$w: $res" | tee -a $out/summary.txt
  case "$res" in *"Terminated"*|"") echo "no result: stopping"; exit 1;; esac
done; done
exit 0

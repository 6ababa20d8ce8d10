# C5 trunk shapes under the channel-slice switches of the raster-window kernels
out=gpurun_out/$1; mkdir -p $out
for s1 in 32 64 128; do RF_CONV_SLICE=$s1 timeout -k 10 300 python tools/trunk_shapes.py bf16 252 448 2> /dev/null | grep "transition1.0.0" | sed "s/^/s1=$s1 /"; done
for s2 in 32 64; do RF_CONV_S2_SLICE=$s2 timeout -k 10 300 python tools/trunk_shapes.py bf16 252 448 2> /dev/null | grep " conv2\|transition1.1\|64  128 3 2" | sed "s/^/s2=$s2 /"; done

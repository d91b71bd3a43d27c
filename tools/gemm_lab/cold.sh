#!/bin/bash
# what makes a launch slower inside the network than back to back?  LAB_COLD modes of lab.hip (see there) on the layer-3 / layer-4
# shapes, without and with the BatchNorm statistics epilogue
L=tools/gemm_lab/lab
for st in 0 1; do
for shape in "50176 512 1024" "50176 1024 512" "12544 2048 1024" "12544 1024 2048"; do
  for c in 0 1 3 5; do
    echo -n "stats=$st LAB_COLD=$c  "
    LAB_COLD=$c $L w7b $shape 20 0 $st | grep -v "tiles_m\|stats rel"
  done
done
done

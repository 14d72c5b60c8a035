#!/bin/bash
# A/B: child blocks in flight per wave in the tree fronts' assembly (UC = 4 product, 5, 9)
for rep in 1 2; do
for v in base uc5 uc9; do
  if [ $v = base ]; then unset G2OHIP_LIB; else export G2OHIP_LIB=$PWD/variants/$v/libg2ohip.so; fi
  echo -n "$v "; bash tools/gpu_ab.sh base
done; done > gpurun_out/r5p_uc.txt 2>&1
cat gpurun_out/r5p_uc.txt

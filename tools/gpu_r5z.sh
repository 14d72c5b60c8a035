#!/bin/bash
# A/B: scheduling strategy of the compiler (variant libraries) against the product build
for rep in 1 2; do
for v in base $VARIANTS; do
  if [ $v = base ]; then unset G2OHIP_LIB; else export G2OHIP_LIB=$PWD/variants/$v/libg2ohip.so; fi
  echo -n "$v "; bash tools/gpu_ab.sh base
done; done 2>&1 | tee gpurun_out/r5z_sched.txt

#!/bin/bash
for v in tfstamps tfstamps300; do
echo "== $v"
G2OHIP_LIB=$PWD/variants/$v/libg2ohip.so G2OHIP_TF_STAMPS_PRINT=1 python bench.py --steps 2 --warmup 3 --no-cpu-baseline --graph off 2>&1 | grep "tree_factor group" | tail -8
done

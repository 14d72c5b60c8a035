#!/bin/bash
# look-ahead pivot block in the band chain kernel (variant library) against the product build: tests, then the band-chain slot
G2OHIP_LIB=$PWD/variants/lookahead/libg2ohip.so python -m pytest tests/test_gpu_band_chain.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
python -m pytest tests/test_gpu_band_chain.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2 3; do
for v in base lookahead; do
  if [ $v = base ]; then unset G2OHIP_LIB; else export G2OHIP_LIB=$PWD/variants/$v/libg2ohip.so; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernels']
print('$v:', round(d['value'],4), 'band chains', round(k['chol_factor(band chains)']['avg_ms'],4), 'tree', round(k['chol_factor(all levels)']['avg_ms'],4), d['residual_rel'])"
done; done 2>&1 | tee gpurun_out/r5la_ab.txt

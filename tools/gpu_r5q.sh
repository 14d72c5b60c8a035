#!/bin/bash
# emulated rank of N: nested-dissection leaf size (chain length against tree height); G2OHIP_ND_LEAF overrides the chain-slot rule
for n in 8 4; do for nl in 0 20 16; do
  if [ $nl = 0 ]; then unset G2OHIP_ND_LEAF; else export G2OHIP_ND_LEAF=$nl; fi
  timeout 300 python bench.py --emulate 0/$n --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5q_${n}_${nl}.log 2>&1
  python - $n $nl <<EOP
import json, sys
try:
    d=json.loads([l for l in open("gpurun_out/r5q_%s_%s.log" % (sys.argv[1], sys.argv[2])) if l.startswith("{")][-1])
    print("N", sys.argv[1], "nd_leaf", sys.argv[2], round(d["value"],4), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if "chol" in k}, {k:d["solver_stats"][k] for k in ("choleskyNNZ","numLevels","bandChains","numFronts")})
except Exception as e:
    print(sys.argv, "failed", e); print(open("gpurun_out/r5q_%s_%s.log" % (sys.argv[1], sys.argv[2])).read()[-800:])
EOP
done; done 2>&1 | tee gpurun_out/r5q_ndleaf_emulated.txt

for nl in ${NLS:-32 48 96}; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --nd-leaf $nl > gpurun_out/ndleaf_$nl.log 2>&1
  python - $nl <<EOP
import json, sys
try:
    d=json.loads([l for l in open("gpurun_out/ndleaf_%s.log" % sys.argv[1]) if l.startswith("{")][-1])
    print("nd_leaf", sys.argv[1], round(d["value"],4), {k:round(v["avg_ms"],4) for k,v in d["kernels"].items() if "chol" in k}, d["solver_stats"], d["residual_rel"])
except Exception as e:
    print(sys.argv[1], "failed", e); print(open("gpurun_out/ndleaf_%s.log" % sys.argv[1]).read()[-800:])
EOP
done

#!/bin/bash
# bench.py on the grid workload (visibility by distance): 10 000 cameras with the CPU oracle beside it, 50 000 without -> gpurun_out/r6_grid.jsonl
: > gpurun_out/r6_grid.jsonl
timeout 1500 python bench.py --workload grid --poses 10000 --steps 10 --warmup 3 2> gpurun_out/grid_10000.err >> gpurun_out/r6_grid.jsonl
timeout 900 python bench.py --workload grid --poses 50000 --steps 5 --warmup 2 --no-cpu-baseline 2> gpurun_out/grid_50000.err >> gpurun_out/r6_grid.jsonl
python - <<EOP
import json
for l in open("gpurun_out/r6_grid.jsonl"):
    d = json.loads(l)
    print(d["config"]["poses"], d["config"]["edges"], round(d["value"], 3), "ms", {k: round(v["avg_ms"], 3) for k, v in d["kernels"].items()}, d["roofline"]["bound"], round(d["roofline"]["achieved"], 2), round(d["roofline"]["frac"], 3), d["residual_rel"], d["solver_stats"]["choleskyNNZ"], d["solver_stats"]["maxFrontDim"], d.get("cpu_baseline", {}).get("value"), d.get("dx_rel_err"), d.get("chi2_rel_err"))
EOP
tail -2 gpurun_out/grid_50000.err
